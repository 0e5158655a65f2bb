"""The store-data hazard scan of a built libscade_hip.so (see tools/check_store_hazard.py for the story): every
> 64-bit buffer / global / flat store of every gfx950 code object in the library must keep its data registers untouched
by the VALU over the next two wait states.  ``scade_amd.build`` runs it behind every link (when llvm-objdump is there)."""
import os, re, struct, subprocess, sys, tempfile

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
                parts = t.split(None, 1)
                if len(parts) == 2:                     # (v_nop and friends: no operands, one wait state)
                    ops_ = [o.strip() for o in parts[1].split(",")]
                    # the destination is the first operand; v_swap_b32 / v_permlane*_swap write BOTH of theirs
                    dsts = ops_[:2] if parts[0].startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) else ops_[:1]
                    if any(regs(o) & d for o in dsts):
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


