import sys; sys.path.insert(0, '/root/repo')
import kallisto_amd as ka
ctx = ka.Context(0)
for nt in (512, 1024):
    for touch in (False, True):
        for rep in range(2):
            print(nt, touch, ctx.grid_barrier(nt, 2000, touch), flush=True)
