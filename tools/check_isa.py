#!/usr/bin/env python3
"""ISA guard for libffhip.so.

On MI355X a packed-fp32 VALU instruction whose LOW result takes the HIGH register of its second source -- v_pk_{add,mul,fma}_f32
... op_sel:[0,1] -- returns a wrong low half in lanes 48-63 while ANOTHER wave of the same SIMD issues v_mfma_f32_16x16x32_f16
(measured: tools/dev/pk_probe.py, profiles/r02_pk_probe.txt).  The engine overlaps one batch's convolution / decode kernels with
the other batch's recurrent layers, and two workgroups of a layer kernel share a CU at H <= 256, so no kernel of the library may
contain that form.  This script disassembles the gfx950 code objects of the built library and fails if one does (the probe
kernel, which issues every form on purpose, is exempt).

usage: tools/check_isa.py [path/to/libffhip.so]        exit status 0 = clean"""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# low result = source 0 low half (op) source 1 HIGH half: the one failing form of the 16 (+ the same with a third source); every
# other op_sel / op_sel_hi combination, v_pk_mov_b32 and a high-half third source of v_pk_fma_f32 were measured clean
BAD = re.compile(r"^\s*v_pk_(?:add|mul|fma)_f32\b.*op_sel:\[0,1")
EXEMPT = ("k_pk_probe",)


def scan(lib, exempt=EXEMPT):
    tmp = tempfile.mkdtemp(prefix="ffhip_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not cos:
            raise SystemExit("no gfx950 code object found in %s" % lib)
        bad, nkern, ninst = [], 0, 0
        for co in cos:
            out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
            kern = None
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    kern = m.group(1); nkern += 1
                    continue
                ninst += 1
                if BAD.match(line) and not any(e in (kern or "") for e in exempt):
                    bad.append((kern, line.strip()))
        return bad, nkern, ninst
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "flappie_amd", "libffhip.so")
    bad, nkern, ninst = scan(lib)
    for kern, line in bad[:40]:
        print("FORBIDDEN  %s\n           %s" % (kern, line))
    print("%s: %d functions, %d lines of disassembly, %d forbidden packed-fp32 forms" % (os.path.basename(lib), nkern, ninst, len(bad)))
    sys.exit(1 if bad else 0)
