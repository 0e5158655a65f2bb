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
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
STORE = re.compile(r"^(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def code_objects(lib):
    """the gfx950 ELF of every offload bundle in the library"""
    b = open(lib, "rb").read()
    out, i = [], b.find(MAGIC)
    while i >= 0:
        n = struct.unpack_from("<Q", b, i + 24)[0]
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, o)
            o += 24
            triple = b[o:o + tl].decode()
            o += tl
            if "gfx950" in triple and size:
                out.append(b[i + off:i + off + size])
        i = b.find(MAGIC, i + 1)
    return out


def disassemble(elf):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        r = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    return r.stdout


def regs(tok):
    m = VREG.fullmatch(tok.strip())
    if not m:
        return set()
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}


def data_regs(op, rest):
    toks = [t.strip() for t in rest.split(",")]
    # buffer_store: vdata, vaddr, srsrc, soffset ; global/flat_store: vaddr, vdata, ...
    return regs(toks[0] if op.startswith("buffer") else toks[1])


def scan(asm, window):
    """[(kernel, store, clobbering instruction, wait states between)]"""
    bad, kernel = [], "?"
    lines = [ln.split("//")[0].strip() for ln in asm.splitlines()]
    for i, ln in enumerate(lines):
        k = re.match(r"^[0-9a-f]+ <(\w+)>:$", ln)
        if k:
            kernel = k.group(1)
        m = STORE.match(ln)
        if not m:
            continue
        d = data_regs(m.group(1), m.group(2))
        ws, j = 0, i + 1
        while ws < window and j < len(lines):
            t = lines[j]
            j += 1
            if not t or t.endswith(":"):
                continue
            if t.startswith(("s_endpgm", "s_branch", "s_setpc")):
                break                                   # (what follows is padding or another block)
            n = re.match(r"s_nop (\d+)", t)
            if n:
                ws += int(n.group(1)) + 1
                continue
            if t.startswith("v_") and not t.startswith(("v_cmp", "v_mfma", "v_smfma")):
                dst = t.split(None, 1)[1].split(",")[0]
                if regs(dst) & d:
                    bad.append((kernel, ln, t, ws))
            ws += 1
    return bad


def check(lib, window=2):
    bad, kernels = [], 0
    for elf in code_objects(lib):
        asm = disassemble(elf)
        kernels += len(re.findall(r"^[0-9a-f]+ <\w+>:$", asm, re.M))
        bad += scan(asm, window)
    return bad, kernels


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
