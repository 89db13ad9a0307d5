"""Disassembly of the gfx950 code object inside a built libmatfact_hip.so (llvm-objdump from /opt/rocm/lib/llvm/bin).

Used by tests/test_isa.py (the hand-written ISA invariants of the kernels, checked on the CPU) and as a command:
    python tools/isa.py [lib.so]                 per-kernel instruction census
    python tools/isa.py [lib.so] <symbol part>   the instructions of every kernel whose demangled name contains the part
Nothing here is on the product path."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "recommender-system_amd", "csrc", "libmatfact_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    return p if os.path.exists(p) else shutil.which(name)


def have_tools():
    return _tool("llvm-objdump") is not None and _tool("llvm-readelf") is not None and shutil.which("c++filt") is not None


def _extract(lib, tmp):
    """path of the gfx950 code object of `lib`, extracted into the directory `tmp`"""
    local = os.path.join(tmp, "lib.so")
    shutil.copy(lib, local)   # --offloading writes the extracted bundles beside its input
    subprocess.check_call([_tool("llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    objs = [f for f in os.listdir(tmp) if "gfx950" in f]
    if not objs:
        raise RuntimeError("no gfx950 code object in " + lib)
    return os.path.join(tmp, objs[0])


def _demangle(names):
    return subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()


def metadata(lib=DEFAULT_LIB):
    """{demangled kernel name: {'.vgpr_count': 34, '.private_segment_fixed_size': 0, ...}} from the code object's notes"""
    with tempfile.TemporaryDirectory() as tmp:
        text = subprocess.check_output([_tool("llvm-readelf"), "--notes", _extract(lib, tmp)], text=True)
    kernels, cur = [], None
    for line in text.splitlines():
        m = re.match(r"^\s+(-\s)?(\.[a-z_]+):\s+(\S+)\s*$", line)
        if line.startswith("  - .") and m:
            cur = {}
            kernels.append(cur)
        if m and cur is not None and len(line) - len(line.lstrip()) <= 4:
            v = m.group(3)
            cur[m.group(2)] = int(v) if re.fullmatch(r"-?\d+", v) else v
    dem = _demangle([k.get(".name", "?") for k in kernels])
    return dict(zip(dem, kernels))


def disassemble(lib=DEFAULT_LIB):
    """{demangled kernel name: [instruction text, ...]} of the gfx950 code object bundled in `lib`."""
    with tempfile.TemporaryDirectory() as tmp:
        text = subprocess.check_output([_tool("llvm-objdump"), "-d", "--mcpu=gfx950", _extract(lib, tmp)], text=True)
    names, bodies, cur = [], [], None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            names.append(m.group(1))
            cur = []
            bodies.append(cur)
            continue
        if cur is None:
            continue
        ins = line.split("//")[0].strip()
        if ins:
            cur.append(ins)
    return dict(zip(_demangle(names), bodies))


_REG = re.compile(r"\b([vsa])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(operand_text, kind="v"):
    """set of register numbers of one kind named in an operand string: 'v[6:9], v12' -> {6, 7, 8, 9, 12}"""
    out = set()
    for k, one, lo, hi in _REG.findall(operand_text):
        if k != kind:
            continue
        if one:
            out.add(int(one))
        else:
            out.update(range(int(lo), int(hi) + 1))
    return out


def split(ins):
    """('v_add_f64', 'v[0:1], v[2:3], v[4:5]')"""
    parts = ins.split(None, 1)
    return parts[0], (parts[1] if len(parts) > 1 else "")


def census(body):
    c = {}
    for ins in body:
        op = split(ins)[0]
        c[op] = c.get(op, 0) + 1
    return c


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = DEFAULT_LIB
    if args and args[0].endswith(".so"):
        lib = args.pop(0)
    kernels = disassemble(lib)
    if args:
        for name, body in kernels.items():
            if args[0] in name:
                print("//", name, len(body), "instructions")
                print("\n".join(body))
    else:
        for name, body in kernels.items():
            if "rocprim" in name:
                continue
            c = census(body)
            pick = {k: v for k, v in c.items() if k.startswith(("v_fma", "v_fmac", "v_mfma", "scratch_", "global_load_lds", "ds_read_b128", "buffer_"))}
            print("%-70s %6d  %s" % (name[:70], len(body), pick))
