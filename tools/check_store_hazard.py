#!/usr/bin/env python3
"""Scan the gfx950 code of every kernel in libscade_hip.so for the store-data hazard LLVM's recognizer exempts.

A buffer / global store of more than 64 bits reads its data registers over several cycles; a VALU write of one
of them in the next issue slots can reach memory.  GCNHazardRecognizer::createsVALUHazard requires one wait state
only when the MUBUF store has NO SGPR soffset; on this part a `buffer_store_dwordx4 ... sN offen nt` followed at
once by `v_add_u32 v<data0>` stored the new value in lanes 12..15 of each 16 (round 5, SaveRiderH of
mlp_tile_f16.h: 1 % gradient error on launches of more than 256 workgroups).  This script pulls the gfx950 code
objects out of the built library (the clang offload bundles of its .hip_fatbin section), disassembles them with
llvm-objdump and reports every >64-bit store whose data registers are written by a VALU instruction within WINDOW
wait states.  Exit status 1 if any.  tests/test_build_cpu.py runs it on every build.

    python tools/check_store_hazard.py [--window 2] [--lib path]
"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scade_amd.build_checks import OBJDUMP, check, scan, code_objects, disassemble  # noqa: E402,F401


def main():
    window = int(sys.argv[sys.argv.index("--window") + 1]) if "--window" in sys.argv else 2
    lib = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else os.path.join(ROOT, "scade_amd", "lib", "libscade_hip.so")
    bad, kernels = check(lib, window)
    print(f"{lib}: {kernels} functions, {len(bad)} store(s) with a VALU write of the data registers within {window} wait states")
    for k, st, nx, ws in bad:
        print(f"   {k[:72]}: {st}   ->  +{ws}: {nx}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
