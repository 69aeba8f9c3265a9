#!/bin/bash
# MFMA utilisation of the prefill kernels at the CURRENT kernel fingerprint (VERDICT r05 item 5 / missing 5): SQ counters, two own
# --pmc passes (kernel-trace only, as gpurun requires).  4096 x 11008 g128, M = 16384 (BASELINE configs[2]) + configs[2]'s attention.
# Usage: pmc_mfma_r06.sh <git head> <kernel fingerprint>   (prints the summary on stdout)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/pmc_mfma_r06
cat > /tmp/r06_prefill.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from autoawq_amd import ops
from bench import rand_packed, rand_packed_nk
dev = torch.device("cuda"); gen = torch.Generator(device=dev).manual_seed(0)
K, N, M = 4096, 11008, 16384
qw, qz, sc = rand_packed(K, N, 128, dev, gen)
x = torch.randn((M, K), device=dev, generator=gen).half()
for kern, nlog in ((ops.KERNEL_REGB, 1), (ops.KERNEL_REGB, 2)):
    for _ in range(3):
        ops.gemm_forward(x, qw, sc, qz, flags=ops.gemm_flags(kern, nlog=nlog))
W = ops.dequantize_weights(qw, sc, qz)
for _ in range(3):
    torch.matmul(x, W)
del W
fq, fz, fs = rand_packed_nk(K, N, 128, dev, gen, fast=True)
for _ in range(3):
    ops.gemv_fast_prefill(x, fq, fs, fz, 128)
# configs[2]'s attention: B 8 x S 2048 x 32 heads x 128
B, S, H, D = 8, 2048, 32, 128
q = torch.randn((B, S, H, D), device=dev, generator=gen).half()
k = torch.randn((B, S, H, D), device=dev, generator=gen).half()
v = torch.randn((B, S, H, D), device=dev, generator=gen).half()
try:
    for _ in range(3):
        ops.prefill_attention(q, k, v, 0)
except Exception as e:  # (signature differences are not worth failing the counter pass)
    print("prefill_attention skipped:", e)
# batched decode, 4096 x 11008, M = 32 on the GEMV layout
nq, nz, ns = rand_packed_nk(K, N, 128, dev, gen)
x32 = x[:32].contiguous()
for _ in range(3):
    ops.gemv_forward(x32, nq, ns, nz, 128)
torch.cuda.synchronize()
PY
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_mfma_r06 -o p1 -- python /tmp/r06_prefill.py > $O/r06_pmc_mfma_1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mfma_r06 -o p2 -- python /tmp/r06_prefill.py > $O/r06_pmc_mfma_2.log 2>&1
echo "# git head $1; kernel source fingerprint: $2"
echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace   (pass 1)"
echo "# rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace   (pass 2)"
echo "# tools/pmc_mfma_r06.sh: 4096 x 11008 g128, M = 16384 (BASELINE configs[2]): gemm_regb 128- and 256-row tiles, the vendor GEMM behind awq_dequant_kernel,"
echo "# the GEMVFast route (repack + gemm_regb FZ form), configs[2]'s attention (B 8 x S 2048 x 32 x 128), gemv_batch at M = 32; mean of 3 dispatches."
echo "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   (GRBM_GUI_ACTIVE is summed over the 8 XCDs)"
grep -h "skipped" $O/r06_pmc_mfma_1.log | sed 's/^/# /'
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_mfma_r06/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:64]
        if any(t in k for t in ("regb", "Cijk", "prefill_attn", "gemv_batch", "repack")):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else float("nan")
    print(f"{k:64s} MFMA busy {100 * busy:5.1f} %  " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(m.items())))
PY
rm -rf $O/pmc_mfma_r06
