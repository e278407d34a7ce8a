"""Reads the phase clocks k_em_sell<true> wrote (KAMD_EM_CLK=<file>): per wavefront of every group, one round of the first chunk.
words: 0 round start, 1 rows pass done, 2 behind barrier 1, 3 columns pass done, 4 behind barrier 2, 5 nrs | ncs << 16, 6 nru | ncu << 32,
7 wall clock (100 MHz) ticks of the round."""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
ng, nw, words, block = (int(x) for x in raw[:4])
c = raw[4:].reshape(ng, nw, words)
waves = block // 64
c = c[:, :waves]
ok = c[:, 0, 0] != 0
c = c[ok]
print(f"groups {ng} (with clocks {ok.sum()}), wavefronts per group {waves}")
rows = c[:, :, 1] - c[:, :, 0]; b1 = c[:, :, 2] - c[:, :, 1]; cols = c[:, :, 3] - c[:, :, 2]; b2 = c[:, :, 4] - c[:, :, 3]
tot = c[:, :, 4].max(1) - c[:, :, 0].min(1)
nrs = c[:, 0, 5] & 0xFFFF; ncs = (c[:, 0, 5] >> 16) & 0xFFFF
nru = c[:, 0, 6] & 0xFFFFFFFF; ncu = (c[:, 0, 6] >> 32) & 0xFFFFFFFF
wall = c[:, 0, 7]
def q(x): return f"mean {x.mean():9.1f}  p10 {np.percentile(x,10):8.0f}  p50 {np.percentile(x,50):8.0f}  p90 {np.percentile(x,90):8.0f}  max {x.max():8.0f}"
print("round, shader clocks      ", q(tot))
print("round, wall ticks (10 ns) ", q(wall))
print("rows pass, slowest wave   ", q(rows.max(1)))
print("rows pass, mean wave      ", q(rows.mean(1)))
print("barrier 1, shortest wait  ", q(b1.min(1)))
print("cols pass, slowest wave   ", q(cols.max(1)))
print("cols pass, mean wave      ", q(cols.mean(1)))
print("barrier 2, shortest wait  ", q(b2.min(1)))
print("row slices / group        ", q(nrs)); print("col slices / group        ", q(ncs))
print("row stream u16 / group    ", q(nru)); print("col stream u16 / group    ", q(ncu))
print("waves with a row slice    ", q(np.minimum(nrs, waves))); print("waves with a col slice    ", q(np.minimum(ncs, waves)))
if words >= 16:
    # first slice of every wavefront: 8 start, 9 after the sums, 10 width | meta << 16, 11 finished (columns); 12..15 the same for rows
    for name, a in (("cols", 8), ("rows", 12)):
        if name == "rows":
            st, sm, wd, fn = c[:, :, 12], c[:, :, 13], c[:, :, 14], c[:, :, 15]
        else:
            st, sm, wd, fn = c[:, :, 8], c[:, :, 9], c[:, :, 10], c[:, :, 11]
        has = st != 0
        width = (wd & 0xFFFF)[has]; meta = ((wd >> 16) & 1)[has]
        t_sum = (sm - st)[has]; t_fin = (fn - sm)[has]
        print(f"{name}: first slices {has.sum()}, with split lanes {meta.mean():.2f}")
        for lo, hi in ((1, 4), (5, 8), (9, 16), (17, 24), (25, 32), (33, 64)):
            sel = (width >= lo) & (width <= hi)
            if sel.any():
                print(f"  width {lo:2d}-{hi:2d}: n {sel.sum():6d}  sums {t_sum[sel].mean():8.1f} clk ({(t_sum[sel] / np.maximum(width[sel], 1)).mean():6.1f} / trip)  scan+finish {t_fin[sel].mean():8.1f}"
                      f"  (split {t_fin[sel & (meta == 1)].mean() if (sel & (meta == 1)).any() else 0:8.1f}, plain {t_fin[sel & (meta == 0)].mean() if (sel & (meta == 0)).any() else 0:8.1f})")
