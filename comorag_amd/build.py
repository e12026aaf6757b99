"""Build libcomorag_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m comorag_amd.build [--force]

Steps
 1. hipcc -c scan_kernels.hip with -save-temps → object + gfx950 assembly
 2. ISA audit of every inline-asm load-ring variant of the scan kernel (see `audit_ring`):
    a variant whose ring registers are touched by compiler-generated code (after their first
    asm load) is marked unsafe
    and the library falls back to the compiler-counted ring for it → ring_audit.cpp
 3. hipcc -c aux_kernels.hip api.hip ring_audit.cpp ; link → comorag_amd/lib/libcomorag_hip.so
The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcomorag_hip.so")
STAMP = os.path.join(LIBDIR, "build_stamp.json")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("CMR_EXTRA_HIPCC_FLAGS", "").split()
if os.environ.get("CMR_BUILD_LIB"):          # experiment builds go to their own file (load with COMORAG_HIP_LIB=...)
    LIB = os.environ["CMR_BUILD_LIB"]
    STAMP = LIB + ".stamp.json"
# The wide kernel's panel loop (3 groups x 4 quads x 8 MFMAs with the epilogue pieces between them) must be unrolled
# completely — every register index is a constant only then; with the slow path inlined at two places of it, its
# unrolled size exceeds LLVM's default limit for "#pragma unroll" (16 K) and hipcc silently keeps the loops.
SCAN_FLAGS = ["-mllvm", "-pragma-unroll-threshold=1048576"]
SOURCES = ["scan_kernels.hip", "aux_kernels.hip", "api.hip", "comm.hip", "ppr.hip", "encoder_kernels.hip", "multi.hip"]
HEADERS = ["cmr_device.h", "cmr_kernels.h", "cmr_internal.h", "cmr_select.h", os.path.join("..", "..", "include", "comorag_hip.h")]

_KERNEL_RE = re.compile(r"^_Z11scan_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEv5ScanP:")


def _regs(text: str) -> set:
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


_LABEL_RE = re.compile(r"^(?:\.L(BB\d+_\d+):|; %bb\.\d+:)")


def _ring_loop_membership(body: list, ring_line: int) -> list:
    """Per line of a kernel body: does it belong to the loop that holds the asm ring loads (line `ring_line` is one of them)?
    LLVM annotates every block with its innermost loop header ('in Loop: Header=BBx_y') and every loop header with its parents
    ('Parent Loop BBx_y'); blocks are laid out in any order (cold blocks of the loop may follow the code behind it)."""
    block_of = [None] * len(body)       # line -> (own label, innermost header, is_header, parents)
    cur = (None, None, False, ())
    parents_of = {}
    n = 0
    while n < len(body):
        m = _LABEL_RE.match(body[n].strip())
        if m:
            label = m.group(1)
            notes = [body[n]]
            j = n + 1
            while j < len(body) and body[j].strip().startswith(";") and not body[j].strip().startswith(";;#") and not _LABEL_RE.match(body[j].strip()):
                notes.append(body[j])
                j += 1
            text = " ".join(notes)
            is_header = "Loop Header" in text
            hm = re.search(r"in Loop: Header=(BB\d+_\d+)", text)
            parents = tuple(re.findall(r"Parent Loop (BB\d+_\d+)", text))
            header = label if is_header else (hm.group(1) if hm else None)
            if is_header and label:
                parents_of[label] = parents
            cur = (label, header, is_header, parents)
        block_of[n] = cur
        n += 1
    if ring_line < 0 or block_of[ring_line] is None or block_of[ring_line][1] is None:
        return [True] * len(body)         # cannot tell: treat everything as inside (the strict rule)
    h = block_of[ring_line][1]
    outer = parents_of.get(h, ())
    main = outer[0] if outer else h        # the outermost loop around the ring loads
    def inside(b):
        if b is None or b[1] is None:
            return False
        return b[1] == main or main in parents_of.get(b[1], ())
    return [inside(b) for b in block_of]


_BRANCH_RE = re.compile(r"^(s_branch|s_cbranch_\w+)\s+\.L(BB\d+_\d+)")
_DRAIN_RE = re.compile(r"^s_waitcnt\s+vmcnt\(0\)")


def _ring_in_flight(body: list, in_loop: list) -> list:
    """Per line: may a load of the asm ring still be in flight there?  Inside the ring loop always (the previous iteration's
    loads).  Outside: on every path that LEAVES the loop, up to the first `s_waitcnt vmcnt(0)` (the compiler's or the kernel's own
    drain statement) — found by walking the control flow (labels, s_branch / s_cbranch, fall-through); code that only runs in
    front of the loop (wherever LLVM placed it in the text) is not reached and may set the registers up."""
    label_at = {}
    for n, l in enumerate(body):
        m = re.match(r"^\.L(BB\d+_\d+):", l.strip())
        if m:
            label_at[m.group(1)] = n
    hot = list(in_loop)
    work = []

    def targets(n):       # (branch targets, falls through?)
        code = body[n].strip().split(";")[0].strip()
        m = _BRANCH_RE.match(code)
        if m:
            return [label_at.get(m.group(2))], m.group(1) != "s_branch"
        if code.startswith("s_endpgm"):
            return [], False
        return [], True

    for n in range(len(body)):
        if not in_loop[n]:
            continue
        tg, fall = targets(n)
        for t in tg:
            if t is not None and not in_loop[t]:
                work.append(t)
        if fall and n + 1 < len(body) and not in_loop[n + 1]:
            work.append(n + 1)
    seen = set()
    while work:
        n = work.pop()
        while n < len(body) and n not in seen and not in_loop[n]:
            seen.add(n)
            code = body[n].strip()
            if _DRAIN_RE.match(code.split(";")[0].strip()):      # (asm statements included: their text is the instruction)
                break
            hot[n] = True
            tg, fall = targets(n)
            for t in tg:
                if t is not None and t not in seen and not in_loop[t]:
                    work.append(t)
            if not fall:
                break
            n += 1
    return hot


def audit_ring(asm_text: str) -> dict:
    """For every scan_kernel<..., ASMRING=1> in the gfx950 assembly decide whether the hand-counted
    load ring is safe: the ring's VGPRs (destinations of the global_load_dwordx4 inside
    ;;#ASMSTART/;;#ASMEND) may be written only by that asm and read only by v_mfma; any other
    compiler instruction touching them after the first ring load (a copy, a spill, a reuse)
    could observe a slot before its data landed (cdna guide §5.7 item 1).  "In flight" is decided on the control flow
    (`_ring_in_flight`): everywhere inside the ring loop, and on the paths that leave it up to the first `s_waitcnt vmcnt(0)`;
    code in front of the loop and behind such a drain may use the registers.  Also requires zero scratch.
    Returns {(dt,nqt,cap,ring,mode): bool}."""
    result = {}
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        m = _KERNEL_RE.match(lines[i])
        if not m:
            i += 1
            continue
        dt, nqt, cap, ring, mode, asmring, _pol = (int(x) for x in m.groups())      # both cache-policy variants of a shape must pass
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            body.append(lines[j])
            j += 1
        i = j
        if not asmring:
            continue
        in_asm = False
        ring_regs: set = set()
        last_ring_load = -1
        for n, l in enumerate(body):
            s = l.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
            elif s.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and s.startswith("global_load_dwordx4"):
                ring_regs |= _regs(s.split(",")[0])
                last_ring_load = n
        ok = len(ring_regs) == 4 * ring
        in_ring_loop = _ring_loop_membership(body, last_ring_load)
        hot = _ring_in_flight(body, in_ring_loop)      # lines at which a ring load may still be in flight
        in_asm = False
        for n, l in enumerate(body):
            s = l.strip()
            if not s or (s.startswith(";") and not s.startswith(";;#ASM")):
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm:
                continue
            if "scratch_" in s:
                ok = False
            code = s.split(";")[0]
            if hot[n] and _regs(code) & ring_regs:
                if code.startswith("v_mfma"):
                    if _regs(code.split(",")[0]) & ring_regs:
                        ok = False
                else:
                    ok = False
        result[(dt, nqt, cap, ring, mode)] = ok and result.get((dt, nqt, cap, ring, mode), True)
    return result


_WIDE_RE = re.compile(r"^_Z16scan_wide_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi0ELi(\d+)EEv5ScanP:")   # ABL = 0 only


def audit_wide(asm_text: str) -> dict:
    """The wide kernel keeps 384 registers of query fragments resident and issues its MFMAs as inline asm (hipcc pads
    no hazards around them).  Per scan_wide_kernel<DT,KS,NT,CAP,NST,KLDS> require: no scratch traffic at all (a spill
    reload inside the panel loop would drain the hand-counted DMA ring), no v_accvgpr_write and no more v_accvgpr_read
    than the epilogue instances account for (anything more means fragments are being shuttled between the register
    files in front of the MFMAs), and no compiler VALU write of an MFMA A/B operand register
    in the three instructions before an asm MFMA (VALU write -> MFMA read needs wait states hipcc does not insert).
    Returns {(dt,ks,nt,cap,waves): problem string or ''}."""
    result = {}
    lines = asm_text.split("\n")
    i = 0
    while i < len(lines):
        m = _WIDE_RE.match(lines[i])
        if not m:
            i += 1
            continue
        dt, ks, nt, cap, nst, klds, nw = (int(x) for x in m.groups())
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            body.append(lines[j].strip())
            j += 1
        i = j
        code = [l.split(";")[0].strip() for l in body if l and not l.startswith(";") and not l.startswith(".")]
        code = [c for c in code if c]
        problems = []
        if any(c.startswith("scratch_") for c in code):
            problems.append("scratch traffic")
        if any(c.startswith("v_accvgpr_write") for c in code):
            problems.append("v_accvgpr_write")
        # Accumulator reads.  An epilogue instance reads a tile's 16 accumulator registers in up to four places: the
        # min / max fold, the (cold) masked fold of the corpus' last panel and the two (cold) candidate pushes (sampling
        # pass / main pass).  NT = 1 has one instance (fold and main-pass push share their reads), the software-pipelined
        # NT = 2 kernel three (tile 0, tile 1 inside the next panel's first quad, tile 1 of the last panel after the
        # loop).  Anything beyond that means query fragments are being shuttled between the register files in front of
        # the MFMAs.
        n_read = sum(c.startswith("v_accvgpr_read") for c in code)
        lim_read = 64 * 3 if nt == 2 else (64 if nw == 8 else 48)      # 8-wave kernel: asm fold + cold partial fold + two pushes that re-read
        if n_read > lim_read:
            problems.append(f"{n_read} v_accvgpr_read > {lim_read}")
        n_mfma = 0
        for n, c in enumerate(code):
            if not c.startswith("v_mfma"):
                continue
            n_mfma += 1
            ops = c.split(",")
            src = _regs(ops[1]) | _regs(ops[2])
            for prev in code[max(0, n - 3):n]:
                if prev.startswith("v_") and not prev.startswith("v_mfma") and not prev.startswith("v_cmp") and _regs(prev.split(",")[0]) & src:
                    problems.append(f"VALU write of an MFMA operand right before it: '{prev}' -> '{c}'")
        if n_mfma < ks * nt:
            problems.append(f"only {n_mfma} MFMAs found")
        result[(dt, ks, nt, cap, nw)] = "; ".join(problems)
    return result


def audit_asm_sgpr_hazard(asm_text: str) -> list:
    """gfx9 needs five wait states between a VALU instruction that writes an SGPR (v_readlane / v_readfirstlane: how hipcc brings a SPILLED
    SGPR back) and a vector-memory instruction that reads that SGPR as its address.  hipcc pads them for its own instructions and not for
    the ones inside an inline-asm statement: the scan kernel's hand-written loads take their base as an "s" operand, and in a variant with
    a hundred spilled SGPRs the reload can sit directly in front of the statement (round 6: the finishing stage's READY hint faulted on a
    stale base in the 8-block-ring variants).  Returns (kernel, line number, writer, asm instruction) for every such pair in the listing —
    the build fails on any.  An `s_nop n` counts n + 1 states, every other instruction one."""
    lines = asm_text.split("\n")

    def is_instr(t: str) -> bool:
        t = t.strip()
        return bool(t) and not t.startswith((";", ".", "//")) and not t.endswith(":")

    found, kern, in_asm = [], None, False
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        if "ASMSTART" in l:
            in_asm = True
            continue
        if "ASMEND" in l:
            in_asm = False
            continue
        if not in_asm or not re.search(r"\b(global|buffer|flat|scratch)_(load|store|atomic)", l):
            continue
        sregs = set()
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", l):
            sregs.update(range(int(a), int(b) + 1))
        sregs.update(int(x) for x in re.findall(r"\bs(\d+)\b", l))
        ws, k = 0, i - 1
        while k >= 0 and ws < 5:
            t = lines[k].strip()
            if is_instr(t):
                w = re.match(r"v_read(?:first)?lane_b32 s(\d+),", t)
                if w and int(w.group(1)) in sregs:
                    found.append((kern, i + 1, t, l.strip()))
                    break
                n = re.match(r"s_nop (\d+)", t)
                ws += int(n.group(1)) + 1 if n else 1
            k -= 1
    return found


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"command failed ({r.returncode}): {' '.join(cmd)}\n{r.stdout[-4000:]}")
    return r.stdout


def _source_hash() -> str:
    h = hashlib.sha256()
    for f in SOURCES + HEADERS + [os.path.join("..", "build.py")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + SCAN_FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    want = _source_hash()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        try:
            if json.load(open(STAMP)).get("hash") == want:
                return LIB
        except Exception:
            pass
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"{HIPCC} not found; cannot build libcomorag_hip.so")
    with tempfile.TemporaryDirectory(prefix="cmr_build_") as tmp:
        objs = []
        # 1. scan kernels with assembly kept
        _run([HIPCC, *FLAGS, *SCAN_FLAGS, "-save-temps", "-c", os.path.join(CSRC, "scan_kernels.hip"), "-o", "scan_kernels.o"], cwd=tmp)
        objs.append(os.path.join(tmp, "scan_kernels.o"))
        asm_file = os.path.join(tmp, f"scan_kernels-hip-amdgcn-amd-amdhsa-{ARCH}.s")
        audit = audit_ring(open(asm_file).read())
        if not audit:       # the mangled-name pattern no longer matches: every variant would silently lose its ring
            raise RuntimeError("ISA audit found no scan_kernel<..., ASMRING=1> in the assembly; update _KERNEL_RE")
        wide = audit_wide(open(asm_file).read())
        if not wide:
            raise RuntimeError("ISA audit found no scan_wide_kernel in the assembly; update _WIDE_RE")
        bad = {k: v for k, v in wide.items() if v}
        if bad:             # there is no second implementation to fall back to: refuse to ship a wide kernel that spills
            raise RuntimeError(f"wide-kernel ISA audit failed: {bad}")
        haz = audit_asm_sgpr_hazard(open(asm_file).read())
        if haz:             # a stale address is a memory fault or — worse — a read of something else
            raise RuntimeError(f"inline-asm vector-memory instruction reads an SGPR a VALU instruction wrote < 5 wait states before: {haz[:4]}")
        # 2. audit table
        rows = ",\n".join(f"    {{{dt}, {nqt}, {cap}, {ring}, {mode}, {1 if ok else 0}}}"
                          for (dt, nqt, cap, ring, mode), ok in sorted(audit.items()))
        with open(os.path.join(tmp, "ring_audit.cpp"), "w") as f:
            f.write("// generated by comorag_amd/build.py from the gfx950 assembly of scan_kernels.hip\n"
                    "struct Row { int dt, nqt, cap, ring, mode, ok; };\n"
                    f"static const Row kRows[] = {{\n{rows}\n}};\n"
                    "// mode: 0 = top-k, 1 = all scores (no candidate lists: cap does not matter), 2 = top-k with the finishing stage\n"
                    "bool cmr_ring_audit_ok(int dtype, int nqt, int cap, int ring, int mode) {\n"
                    "    bool any = false;\n"
                    "    for (const Row& r : kRows)\n"
                    "        if (r.dt == dtype && r.nqt == nqt && r.ring == ring && r.mode == mode && (r.mode == 1 || r.cap == cap)) {\n"
                    "            if (!r.ok) return false;\n"
                    "            any = true;\n"
                    "        }\n"
                    "    return any;\n"
                    "}\n")
        for src in ("aux_kernels.hip", "api.hip", "comm.hip", "ppr.hip", "encoder_kernels.hip", "multi.hip"):
            o = os.path.join(tmp, src.replace(".hip", ".o"))
            _run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", o], cwd=tmp)
            objs.append(o)
        o = os.path.join(tmp, "ring_audit.o")
        _run([HIPCC, "-O2", "-std=c++17", "-fPIC", "-c", os.path.join(tmp, "ring_audit.cpp"), "-o", o], cwd=tmp)
        objs.append(o)
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    n_ok = sum(audit.values())
    info = {"hash": want, "arch": ARCH, "wide_variants_audited": len(wide), "asm_ring_variants": len(audit), "asm_ring_safe": n_ok,
            "unsafe": [list(k) for k, v in sorted(audit.items()) if not v]}
    json.dump(info, open(STAMP, "w"), indent=1)
    if verbose:
        print(f"[comorag_amd.build] built {LIB}; asm-ring variants safe {n_ok}/{len(audit)}; unsafe: {info['unsafe']}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
