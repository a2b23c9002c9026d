"""Static scan of the device code inside libxm3d_hip.so: per kernel the instruction count, AGPR copies (v_accvgpr_*), compare / select
cascades (v_cmp_eq_u32: hipcc's lowering of a vector extract with a run-time index), scratch accesses (register spills) and narrow (<= 4-byte)
against wide global loads.  This is how
the round-4 attention and GEGLU findings were made; tests/test_isa_scan.py keeps them fixed.
usage: python tools/isa_scan.py [path/to/libxm3d_hip.so]"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path, arch="gfx950"):
    """the device code objects of `arch` embedded in a hipcc-built shared library (clang offload bundles in .hip_fatbin)"""
    data = open(so_path, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        off = i + len(MAGIC)
        (num,) = struct.unpack_from("<Q", data, off)
        off += 8
        for _ in range(num):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if arch in triple and sz > 0:
                out.append(data[i + o:i + o + sz])
        pos = i + len(MAGIC)


def scan(so_path=None):
    """{demangled-ish kernel symbol: Counter(total, accvgpr, cmp_eq, scratch, mfma)}"""
    so_path = so_path or os.path.join(ROOT, "xmask3d_amd", "libxm3d_hip.so")
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n, co in enumerate(code_objects(so_path)):
            p = os.path.join(tmp, f"co_{n}.co")
            with open(p, "wb") as f:
                f.write(co)
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", p], check=True, capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = res.setdefault(m.group(1), collections.Counter())
                    continue
                if cur is None or not line.startswith("\t"):
                    continue
                cur["total"] += 1
                if "v_accvgpr" in line:
                    cur["accvgpr"] += 1
                elif "v_cmp_eq_u32" in line:
                    cur["cmp_eq"] += 1
                elif "scratch_" in line:
                    cur["scratch"] += 1
                elif "v_mfma" in line:
                    cur["mfma"] += 1
                else:
                    op = line.split()[0]
                    if op in ("global_load_dword", "global_load_ushort", "global_load_ubyte", "global_load_sshort", "global_load_short_d16"):
                        cur["narrow_loads"] += 1  # <= 4 bytes per lane
                    elif op.startswith("global_load_dwordx"):
                        cur["wide_loads"] += 1
    return res


def short(sym):
    """xm3d kernel name with its integer template arguments, from the mangled symbol"""
    m = re.search(r"xm3d\d+(k_[a-z0-9_]+?)I((?:L[ib]\d+E)+)E", sym)
    if m:
        return m.group(1) + "<" + ",".join(re.findall(r"L[ib](\d+)E", m.group(2))) + ">"
    m = re.search(r"xm3d\d+(k_[a-z0-9_]+)", sym)
    return m.group(1) if m else sym[:60]


if __name__ == "__main__":
    r = scan(sys.argv[1] if len(sys.argv) > 1 else None)
    rows = [(c["cmp_eq"], c["accvgpr"], c["scratch"], c["total"], c["mfma"], short(k)) for k, c in r.items() if "xm3d" in k]
    rows.sort(key=lambda t: (-(t[0] > 100), -t[1], -t[2], -t[3]))
    print(f"{len(rows)} xm3d kernels\ncmp_eq accvgpr scratch  total  mfma  kernel")
    for t in rows[:60]:
        print("%6d %7d %7d %6d %5d  %s" % t)
