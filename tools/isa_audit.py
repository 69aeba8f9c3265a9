#!/usr/bin/env python3
"""Static audit of the compiled kernels for the store-data hazard of profiles/r01_store_hazard.txt:
a >8-byte MUBUF store whose soffset is an SGPR gets no wait states from the compiler before its data
VGPRs are rewritten, and gfx950 reads them late.  The rule in the sources is "16-byte buffer stores keep
soffset = 0"; this script checks the generated ISA.  Usage: tools/isa_audit.py [file.hip ...]
Exit status 1 if any 12/16-byte buffer store uses an SGPR soffset."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "autoawq_amd", "csrc")
STORE = re.compile(r"\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")


def audit(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-fno-slp-vectorize", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    stores, bad = 0, []
    for i, l in enumerate(lines):
        m = STORE.match(l)
        if not m:
            continue
        stores += 1
        if m.group(5).startswith("s"):
            bad.append((i + 1, l.strip()))
    return os.path.basename(src), stores, bad


def audit_asm_loads(src):
    """The hand-written loads of the register-decoded kernels (gemm_regb.hip, gemm_skinny.hip): every asm block that
    holds a buffer load opens with s_nop 4 (SALU-written SGPR operands need five wait states before a VMEM instruction
    reads them; hipcc pads nothing for asm operands), an LDS-DMA block writes M0 itself and waits before using it, and
    the kernels touch no scratch memory (scratch traffic counts in vmcnt and would break the counted waits).
    Returns (file, blocks seen, list of problems)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-fno-slp-vectorize", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    blocks, bad, cur = 0, [], None
    for i, l in enumerate(lines):
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            cur = []
        elif t.startswith(";;#ASMEND"):
            body = [x for x in cur if x and not x.startswith(";")]
            if any(x.startswith("buffer_load") for x in body):
                blocks += 1
                first_load = next(k for k, x in enumerate(body) if x.startswith("buffer_load"))
                if not any(x.startswith("s_nop 4") for x in body[:first_load]):
                    bad.append((i + 1, "buffer load without a leading s_nop 4: " + " | ".join(body[:3])))
                if any(" lds" in x for x in body) and not body[0].startswith("s_mov_b32 m0"):
                    bad.append((i + 1, "LDS-DMA block does not set M0 itself: " + " | ".join(body[:3])))
            cur = None
        elif cur is not None:
            cur.append(t)
        elif t.startswith("scratch_"):
            bad.append((i + 1, "scratch access: " + t))
    return os.path.basename(src), blocks, bad


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit_inflight_regs(src):
    """The row-streaming GEMV kernel (gemv_rows.hip) requests a ring slot in one asm block (global loads with VGPR
    destinations) and releases it in a later asm block `s_waitcnt vmcnt(N) ; releases <regs>`.  hipcc believes the
    destinations are written when the request block ends, so it MAY copy or reuse them while the loads are in flight
    (cdna_hip_programming.md 5.7, item 1).  This walks every path from each request block to the first wait block that
    names one of its registers (or the vmcnt(0) drain) and fails if (a) that wait does not name exactly the request's
    registers (a compiler copy happened in between) or (b) any instruction on the way touches one of them.
    Returns (file, request blocks checked, list of problems)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm",
                               "-fno-slp-vectorize", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-S",
                               "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out).read().splitlines()]
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^[.\w$]+:", l)}
    # asm blocks: (start, end, kind, regs)
    blocks, i = {}, 0
    while i < len(lines):
        if lines[i].startswith(";;#ASMSTART"):
            j = i + 1
            while not lines[j].startswith(";;#ASMEND"):
                j += 1
            body = lines[i + 1:j]
            loads = [x for x in body if x.startswith("global_load") and " lds" not in x and "_lds_" not in x]
            waits = [x for x in body if x.startswith("s_waitcnt vmcnt")]
            if loads:
                dst = set()
                for x in loads:
                    dst |= regs_of(x.split(",")[0])
                blocks[i] = (j, "request", dst)
            elif waits:
                named = regs_of(waits[0].split(";", 1)[1]) if ";" in waits[0] else set()
                blocks[i] = (j, "drain" if "vmcnt(0)" in waits[0] and not named else "wait", named)
            i = j
        i += 1
    checked, bad = 0, []
    for k, t in enumerate(lines):  # scratch traffic (spills) counts in vmcnt and would break the counted waits
        if t.startswith(("scratch_", "buffer_store_dword v", "buffer_load_dword v")) and "Spill" in t or t.startswith("scratch_"):
            bad.append((k + 1, "scratch access: " + t))
    for start, (end, kind, dst) in blocks.items():
        if kind != "request":
            continue
        checked += 1
        seen, todo = set(), [end + 1]
        while todo:
            k = todo.pop()
            while k < len(lines) and k not in seen:
                seen.add(k)
                t = lines[k]
                if k in blocks:
                    e2, kind2, regs2 = blocks[k]
                    if kind2 == "drain":
                        break
                    if kind2 == "wait" and regs2 & dst:
                        if not dst <= regs2:  # (a wait may release several request blocks at once: a superset is fine)
                            bad.append((k + 1, f"wait names {sorted(regs2)} but the request at line {start + 1} wrote {sorted(dst)}"))
                        break
                    if kind2 == "request" and regs2 & dst:
                        bad.append((k + 1, f"slot requested again before it was released (request at line {start + 1})"))
                        break
                    k = e2 + 1
                    continue
                if t.startswith("s_endpgm"):
                    bad.append((k + 1, f"request at line {start + 1} never released"))
                    break
                if t and not t.startswith((";", ".")) and not re.match(r"^[.\w$]+:", t) and regs_of(t) & dst:
                    bad.append((k + 1, f"touches in-flight registers of the request at line {start + 1}: {t}"))
                m = re.match(r"s_(c?branch)\w*\s+(\S+)", t)
                if m and m.group(2) in labels:
                    todo.append(labels[m.group(2)])
                    if m.group(1) == "branch":
                        break
                k += 1
    return os.path.basename(src), checked, bad


def main(files):
    files = files or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:
        res = list(ex.map(audit, files))
    rc = 0
    for name, stores, bad in res:
        print(f"{name:20s} 12/16-byte buffer stores: {stores:4d}   with an SGPR soffset: {len(bad)}")
        for ln, text in bad[:10]:
            print(f"    line {ln}: {text}")
            rc = 1
    return rc, res





# ------------------------------------------------------------------------------------------------------------------------------
# audit_vmcnt: are the hand-counted `s_waitcnt vmcnt(N)` of the asm-load kernels RIGHT?  (audit_inflight_regs above checks
# that nothing touches a named register set between request and release; it takes the count itself on trust.)
VMEM = re.compile(r"^(buffer_load|buffer_store|buffer_atomic|global_load|global_store|global_atomic|scratch_load|scratch_store|flat_load|flat_store)")


def compile_isa(src, flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm",
                               "-fno-slp-vectorize", "-DAWQ_BUILDING_LIB", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                               "--cuda-device-only", src, "-o", out] + list(flags), stderr=subprocess.DEVNULL)
        return [l.strip() for l in open(out).read().splitlines()]


def audit_vmcnt(src, flags=(), max_states=400000):
    """Simulates the vector-memory queue of every kernel in `src` over its control-flow graph: vector-memory operations retire in
    issue order (loads and stores both count on gfx950), `s_waitcnt vmcnt(N)` retires all but the N youngest.  A violation is any
    instruction that names a VGPR which is the destination of a load still in the queue -- a counted wait that is too lax, a
    compiler copy / spill / reuse of an in-flight register, or a request into a register whose previous load has not landed.
    Both successors of every conditional branch are followed; a (program counter, queue) state is visited once, so loops run to
    their steady state.  The model is validated by the compiler's own waits: every compiler-tracked load of the same kernels must
    come out clean too.  Returns (file, kernels, vector-memory operations seen, list of problems)."""
    lines = compile_isa(src, flags)
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^[.\w$]+:", l)}
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) or (re.match(r"^\w+:", l) and i + 1 < len(lines) and "; @" in l)]
    bad, nops, nk = [], 0, 0
    for k0 in kernels:
        if not any(lines[i].startswith("s_endpgm") for i in range(k0, min(len(lines), k0 + 200000))):
            continue
        nk += 1
        seen, todo, reported = set(), [(k0 + 1, ())], set()
        while todo:
            pc, queue = todo.pop()
            queue = list(queue)
            while pc < len(lines):
                key = (pc, tuple(queue))
                if key in seen:
                    break
                seen.add(key)
                if len(seen) > max_states:
                    bad.append((pc + 1, "state limit reached: audit incomplete"))
                    todo = []
                    break
                t = lines[pc]
                if not t or t[0] in ";." or re.match(r"^[.\w$]+:", t):
                    pc += 1
                    continue
                if t.startswith("s_endpgm"):
                    break
                if t.startswith("s_waitcnt"):
                    m = re.search(r"vmcnt\((\d+)\)", t)
                    if m:
                        n = int(m.group(1))
                        if len(queue) > n:
                            queue = queue[len(queue) - n:] if n else []
                    pc += 1
                    continue
                touched = regs_of(t.split(";")[0])
                is_vmem = bool(VMEM.match(t))
                if is_vmem and "_load" in t.split()[0] and " lds" not in t:
                    # a load INTO a register whose previous load is still in flight is legal (loads return in order: the younger
                    # one lands last) -- compiler-generated polling loops do it; only the address operands must have landed
                    touched -= regs_of(t.split(",")[0])
                inflight = set().union(*[q for q in queue]) if queue else set()
                hit = touched & inflight
                if hit and (pc, tuple(sorted(hit))) not in reported:
                    reported.add((pc, tuple(sorted(hit))))
                    bad.append((pc + 1, f"names v{sorted(hit)} while their load is in flight: {t[:90]}"))
                if is_vmem:
                    nops += 1
                    is_load = "_load" in t.split()[0] or ("atomic" in t.split()[0] and " glc" in t or " sc0" in t and "atomic" in t.split()[0])
                    dst = frozenset()
                    if is_load and " lds" not in t and "_lds_" not in t.split()[0]:
                        dst = frozenset(regs_of(t.split(",")[0]))
                    if dst:  # a register requested again is governed by the YOUNGER load from here on (in-order return)
                        queue = [q - dst for q in queue]
                    queue.append(dst)
                    if len(queue) > 64:  # (vmcnt is a 6-bit counter: the hardware itself stalls the 65th request)
                        queue = queue[-64:]
                    while queue and not queue[0]:  # the oldest entries without a destination register cannot matter any more
                        queue.pop(0)
                m = re.match(r"s_(c?branch)\w*\s+(\S+)", t)
                if m and m.group(2) in labels:
                    todo.append((labels[m.group(2)], tuple(queue)))
                    if m.group(1) == "branch":
                        break
                pc += 1
    return os.path.basename(src), nk, nops, bad


def main_vmcnt(files):
    files = files or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    rc = 0
    for f in files:
        name, kernels, visits, bad = audit_vmcnt(f, max_states=8000000)
        print(f"{name:20s} kernels {kernels:3d}   vector-memory operations visited {visits:8d}   problems {len(bad)}")
        for ln, text in bad[:10]:
            print(f"    line {ln}: {text}")
            rc = 1
    return rc


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--vmcnt":  # tools/isa_audit.py --vmcnt [file.hip ...]
        sys.exit(main_vmcnt(sys.argv[2:]))
    sys.exit(main(sys.argv[1:])[0])
