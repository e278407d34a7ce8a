# where kamd_index_load's time goes on a GPU box's host (config #3's index): KAMD_INDEX_TIMING=1
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench, kallisto_amd as ka
cat, tl, idx = bench.prepare_workload("human", 20000, True)
os.environ["KAMD_INDEX_TIMING"] = "1"
for rep in range(2):
    t = time.time(); ix = ka.Index(idx); print("kamd_index_load", round(time.time() - t, 3), "s", flush=True)
    del ix
PY
