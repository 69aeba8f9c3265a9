import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_moe
from autoawq_amd.modules.fused import moe
for rep in range(2):
    for fuse in (True, False):
        moe.FUSE_ACTIVATION_INTO_W2 = fuse
        r = bench_moe.run(verbose=False)
        print("fuse", fuse, {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k in ("us_per_block", "experts_hit")}, flush=True)
