"""Guard-band runs: the kernels once more, in a child process whose EVERY device allocation sits flush
against unmapped address space (tests/guard/guard_alloc.cpp through torch's pluggable-allocator hook) -- at the END of its own
mapping (an over-read or over-write of one byte past any operand faults) and at the START (one byte before).  torch's caching
allocator packs tensors into shared 2 MiB / 20 MiB segments, where such an access lands in a neighbour and goes unnoticed.
A GPU memory fault aborts the child: the assertion shows the last test it had started."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.guard_skip]

# the kernel-facing tests (every HIP kernel family, BASELINE shapes + ragged ones, M = 1 ... 16 and prefill sizes); the
# whole suite runs under the same allocator by tools/guard_run.sh (profiles/r04_fault_hunt/)
SUBSET = ("test_gemv_rows_kernel_vs_oracle or test_gemv_rows_block_fusions or config4 or dequant or unpack or gemv_lds or gemvfast or "
          "test_gemm_vs_oracle_all_variants or attention or rmsnorm or rope or silu or moe or chain or gemv_batch or repack or "
          "gemvfast_layout_prefill_route or route_a_replay")
# default selection: one placement (flush against the END of the mapping: an over-read past an operand, the usual failure) of a compact
# subset -- one family each; AWQ_FULL_MATRIX=1: both placements of the whole subset (135 s each on an MI355X; round 5: the driver's
# suite is held near three minutes)
# (of the two many-shape kernel tests: the BASELINE shape pair + every ragged / tiny / multi-pass shape -- where an access past an operand
#  would be -- not the other large ones)
COMPACT = ("(test_gemv_rows_kernel_vs_oracle and (4096-4096-128 or 11008-4096-128 or 4096-4099 or 384-7 or 1280-10 or 2048-200 or 256-16)) or "
           "(test_gemv_batch_kernel_vs_oracle and (4096-11008 or 11008-4096 or 2048-4099 or 16512-72 or 256-16 or 1024-200)) or "
           "(test_gemvfast_layout_prefill_route_vs_oracle and (1-1024-200 or 1-512-64 or 1-4096-4096)) or "  # round 6: the FZ form + its repack, ragged N
           "(test_grouped_rows_kernel_vs_oracle and (5-3-512-250 or 4-2-14336-256 or 1-2-1024-96)) or test_grouped_prefill_gather_scatter or "  # round 6: MoE decode twins,
           "test_moe_sort_pairs or test_moe_route_routing_only or "                                                                                # the index-list prefill
           "test_gemv_batch_refuses or repack or prefill_attention or test_decode_attention_vs_oracle or unpack or "
           "test_dequant_golden or test_gemm_golden or test_moe_block_vs_oracle or rmsnorm or rope_kv or test_gemvfast_layout_golden")


def _env(mode):
    env = dict(os.environ)
    env["AWQ_GUARD_ALLOC"] = mode
    return env


def _lib_built():
    from tests.guard import build

    return build.build()


@pytest.mark.gpu_fault
@pytest.mark.parametrize("mode", ["end", "start"])
def test_guard_allocator_faults_on_an_access_outside_an_allocation(mode):
    """The checker itself: a read 8 KiB outside a guarded allocation must kill the child (else the runs below prove nothing).
    Provokes a real GPU memory fault: selected only with AWQ_RUN_FAULT_SELFCHECK=1 (tests/conftest.py); the evidence runs of
    tools/guard_run.sh perform the same self-check (profiles/r04_final_195ff05/guard/selfcheck_end.log: "Memory access fault")."""
    _lib_built()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_selfcheck.py")], env=_env(mode), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert "inside:" in r.stdout, r.stdout + r.stderr
    assert r.returncode != 0 and "must not get here" not in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])


@pytest.mark.parametrize("mode", ["end", "start"])
def test_kernels_touch_nothing_outside_their_operands(mode):
    full = os.environ.get("AWQ_FULL_MATRIX", "") == "1"
    if mode == "start" and not full:
        pytest.skip("the START placement runs with AWQ_FULL_MATRIX=1 (tools/final_r05.sh)")
    _lib_built()
    cmd = [sys.executable, "-X", "faulthandler", "-m", "pytest", "tests", "-m", "gpu", "-v", "-p", "no:cacheprovider", "-x", "-k", SUBSET if full else COMPACT]
    r = subprocess.run(cmd, env=_env(mode), cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-12:])
    assert r.returncode == 0, f"guard-band run ({mode}) rc {r.returncode}:\n{tail}\n{r.stderr[-1500:]}"
    assert "guard-band allocator: {" in r.stdout and f"'placement': '{mode}'" in r.stdout, tail  # the hook really was active
    assert " passed" in tail and "failed" not in tail, tail
