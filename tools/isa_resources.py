"""Register / LDS / scratch use of every scan_kernel and scan_wide_kernel variant, from the gfx950 assembly hipcc emits (no GPU needed), with the
build's ring-audit verdict:  python tools/isa_resources.py > profiles/<round>_isa_resources.txt"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comorag_amd import build as B

with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([B.HIPCC, *B.FLAGS, *B.SCAN_FLAGS, "-save-temps", "-c", os.path.join(B.CSRC, "scan_kernels.hip"), "-o", "s.o"], cwd=tmp, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(tmp, f"scan_kernels-hip-amdgcn-amd-amdhsa-{B.ARCH}.s")).read()
audit = B.audit_ring(asm)
print("# scan_kernel<DT, NQT, CAP, R, MODE, ASMRING, POL>: DT 0 f32 / 1 bf16 / 2 f16; MODE 0 top-k, 1 all scores, 2 top-k with the finishing stage;")
print("# waves/SIMD = floor(512 / vgprs) (unified file); audit = build.py:audit_ring for the ASMRING = 1 variants of the shape")
print(f"{'kernel':58s} {'vgpr':>5s} {'sgpr':>5s} {'sgpr spill':>10s} {'scratch B':>9s} {'waves/SIMD':>10s}  audit")
for m in re.finditer(r"\.name:\s+(_Z\d+scan(?:_wide)?_kernelI(\w+?)EEv5ScanP)\n(.*?)\.wavefront_size", asm, re.S):
    body = m.group(3)
    f = dict(re.findall(r"\.(vgpr_count|sgpr_count|private_segment_fixed_size|sgpr_spill_count|agpr_count):\s+(\d+)", body))
    ints = [int(x) for x in re.findall(r"Li(\d+)E", m.group(2) + "E")]
    wide = "wide" in m.group(1)
    name = ("scan_wide_kernel<" if wide else "scan_kernel<") + ", ".join(map(str, ints)) + ">"
    verdict = ""
    if not wide and len(ints) == 7 and ints[5] == 1:
        verdict = "safe" if audit.get(tuple(ints[:5]), False) else "UNSAFE"
    v = int(f.get("vgpr_count", 0))
    print(f"{name:58s} {v:5d} {int(f.get('sgpr_count', 0)):5d} {int(f.get('sgpr_spill_count', 0)):10d} {int(f.get('private_segment_fixed_size', 0)):9d} {512 // max(v, 1) if v else 0:10d}  {verdict}")
