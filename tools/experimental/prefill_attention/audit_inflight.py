"""CPU check of the -DAWQ_PATTN_ASM_LOADS build of prefill_attn.hip: compiles it to ISA and walks each kernel in layout order from
every asm `buffer_load_dwordx4` to the counted `s_waitcnt vmcnt(N)` that retires it -- no instruction in between (no compiler
copy, no spill, no reuse as a temporary) may name a register whose load is still in flight, and the tile loop must hold exactly
the two counted waits (no compiler-inserted vmcnt).  Same idea as tools/isa_audit.py::audit_inflight_regs for csrc/gemv_rows.hip.

    python tools/experimental/prefill_attention/audit_inflight.py        # exit status 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))


def regs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-DAWQ_PATTN_ASM_LOADS",
                               "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "autoawq_amd", "csrc"),
                               "-S", "--cuda-device-only", "-o", out, os.path.join(HERE, "prefill_attn.hip")], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    bad = 0
    starts = [i for i, l in enumerate(lines) if re.match(r"_ZN.*awq_prefill_attn_kernel.*:", l)]
    for s in starts:
        e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith("s_endpgm"))
        body = [l.strip() for l in lines[s:e]]
        inflight, in_asm, loads, problems, compiler_waits_in_loop = {}, False, 0, [], 0
        loop = [i for i, t in enumerate(body) if "Loop Header: Depth=1" in t]
        loop_lo = loop[0] if loop else 0
        label = re.match(r"\.?(LBB\d+_\d+)", body[loop_lo]).group(1) if loop else None
        loop_hi = max(i for i, t in enumerate(body) if label and label in t) if loop else 0
        for i, t in enumerate(body):
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t[0] in ";.":
                continue
            if in_asm and t.startswith("buffer_load_dwordx4"):
                m = re.match(r"buffer_load_dwordx4 v\[(\d+):(\d+)\]", t)
                for r in range(int(m.group(1)), int(m.group(2)) + 1):
                    inflight[r] = i
                loads += 1
                continue
            if t.startswith("s_waitcnt") and "vmcnt" in t:
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                if not in_asm and loop_lo < i < loop_hi:
                    compiler_waits_in_loop += 1
                issued = sorted(set(inflight.values()))
                keep = set(issued[len(issued) - n:]) if n else set()
                inflight = {r: l for r, l in inflight.items() if l in keep}
                continue
            hit = regs(t) & set(inflight)
            if hit:
                problems.append((i, t, sorted(hit)))
        name = lines[s].split(":")[0]
        print(f"{name}: {loads} asm loads, {len(problems)} touches of in-flight registers, {compiler_waits_in_loop} compiler vmcnt waits inside the tile loop")
        for pr in problems[:5]:
            print("    ", pr)
        bad += len(problems) + compiler_waits_in_loop
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
