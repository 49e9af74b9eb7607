"""Build-time proof for the hand-written operand staging of gemm_encode_f16x3.hip.

The staging is inline asm that sets m0 (the LDS base of `global_load_lds`) itself.  hipcc does not model m0 as clobberable
("reserved register"), so the asm statements do not list it -- which is only sound if the compiler never keeps a value of
its own in m0 across them.  This script compiles the file to gfx950 assembly and checks exactly that:
  * every instruction that mentions m0 is `s_mov_b32 m0, <sgpr>` (ours: two per staging call),
  * the only instructions that read m0 implicitly are `global_load_lds_dwordx4` (ours: four per staging call), and
  * the two counts match (2 loads per m0 write).
Run by `make check-m0` and by __graft_entry__.build().
"""
import os
import pathlib
import re
import shlex
import shutil
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "saev_amd" / "csrc" / "gemm_encode_f16x3.hip"
OUT = ROOT / "build" / "gemm_encode_f16x3.s"
IMPLICIT_M0 = re.compile(r"\b(s_movrel\w*|v_movrel\w*|ds_gws\w*|ds_append|ds_consume|s_sendmsg\w*|buffer_load\w*.*\blds\b|ds_\w*_gs_reg\w*)")


def main() -> int:
    OUT.parent.mkdir(exist_ok=True)
    # the Makefile hands over its own compiler, target and flags (make check-m0), so that the proof runs on the assembly of
    # exactly the build that is linked into libsaev_amd.so; stand-alone runs fall back to the Makefile's defaults
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = shlex.split(os.environ.get("HIPFLAGS", "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-value -Iinclude"))
    if shutil.which(hipcc) is None:
        print(f"check_m0: {hipcc} not found -- skipped (advisory without a compiler)")
        return 0
    cmd = [hipcc, *flags, "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", str(SRC), "-o", str(OUT)]
    subprocess.run(cmd, check=True, cwd=ROOT)
    writes = loads = 0
    bad = []
    for n, line in enumerate(OUT.read_text().splitlines(), 1):
        code = line.split(";")[0].strip()
        if not code or code.startswith((".", "//")) or code.endswith(":"):
            continue
        if "global_load_lds_dwordx4" in code:
            loads += 1
        elif re.search(r"\bm0\b", code):
            if re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)", code):
                writes += 1
            else:
                bad.append((n, code))
        elif IMPLICIT_M0.search(code):
            bad.append((n, code))
    ok = not bad and writes > 0 and loads == 2 * writes
    print(f"check_m0: {writes} m0 writes, {loads} global_load_lds, {len(bad)} foreign m0 uses -> {'ok' if ok else 'FAILED'}")
    for n, code in bad[:20]:
        print(f"  {OUT.name}:{n}: {code}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
