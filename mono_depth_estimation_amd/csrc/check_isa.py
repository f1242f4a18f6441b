#!/usr/bin/env python3
"""Build check: no gfx950 kernel of the library may contain a packed-fp32 VALU instruction with a lane-crossing op_sel
(v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ... op_sel:[..]).

Why: with such an instruction the LOW lane's result (the one op_sel feeds from the HIGH register of
a source pair) came out wrong in a few lanes whenever a wave of ANOTHER kernel shared the SIMD -- the fused BatchNorm-backward
sums of the 8-wave 64-column conv tiles beside a weight-gradient workgroup of the second stream (30 of 30 runs), a sparse
one-ulp variation in BTS' planar-guidance backward.  tools/probes/pk_opsel_probe.hip shows it on the bare instruction: beside an
MFMA loop of another kernel the low lane of v_pk_add / mul / fma_f32 op_sel:[0,1] (source 1 taken from the high register) is
wrong in up to 10 % of the results, and never alone or beside a VALU loop.  The same instructions in natural lane order are fine
(every BatchNorm kernel is full of them), so is op_sel on source 0 and on v_pk_mov_b32; the pattern below is a superset.  hipcc's SLP vectoriser is what emits the op_sel forms (it pairs the odd / even halves of an unpacked
16-bit pair crosswise); build.sh compiles the affected files with -fno-slp-vectorize, and this check keeps a later change
from bringing them back.

    python3 check_isa.py ../libmde_hip.so [more .so]        exit code 1 and the kernels' names if any such instruction is found
"""
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAD = re.compile(r"v_pk_(?:add|mul|fma)_f32\b.*\bop_sel:")


def fatbin(path):
    """The bytes of the .hip_fatbin section."""
    out = subprocess.run([READELF, "-S", "-W", path], capture_output=True, text=True, check=True).stdout
    for line in out.splitlines():
        m = re.search(r"\.hip_fatbin\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
        if m:
            off, size = int(m.group(2), 16), int(m.group(3), 16)
            with open(path, "rb") as f:
                f.seek(off)
                return f.read(size)
    raise SystemExit("%s: no .hip_fatbin section" % path)


def code_objects(blob):
    """Every gfx950 code object of every offload bundle in the section (one bundle per translation unit).  A section that
    holds a COMPRESSED bundle instead (magic CCOB: librccl.so, for one) goes through clang-offload-bundler."""
    if MAGIC not in blob and blob[:4] == b"CCOB":
        with tempfile.TemporaryDirectory() as d:
            src = d + "/fat.bin"
            with open(src, "wb") as f:
                f.write(blob)
            listed = subprocess.run([BUNDLER, "--list", "--type=o", "--input=" + src], capture_output=True, text=True, check=True).stdout
            for i, triple in enumerate(t for t in listed.split() if "gfx950" in t):
                out = "%s/%d.co" % (d, i)
                subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=" + triple, "--input=" + src, "--output=" + out], check=True)
                with open(out, "rb") as f:
                    yield f.read()
        return
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def check(path):
    bad = {}
    nobj = nkern = 0
    for co in code_objects(fatbin(path)):
        nobj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as t:
            t.write(co)
            t.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", t.name], capture_output=True, text=True, check=True).stdout
        kern = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kern = m.group(1)
                nkern += 1
            elif kern and BAD.search(line):
                bad[kern] = bad.get(kern, 0) + 1
    if nobj == 0:
        raise SystemExit("%s: no gfx950 code object found (compressed bundles?)" % path)
    return nobj, nkern, bad


def main(paths):
    rc = 0
    for path in paths:
        nobj, nkern, bad = check(path)
        if bad:
            rc = 1
            print("ERROR: %s: packed-fp32 instructions with op_sel in %d kernel(s):" % (path, len(bad)), file=sys.stderr)
            for k, c in sorted(bad.items()):
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
                print("   %4d  %s" % (c, name[:160]), file=sys.stderr)
        else:
            print("%s: %d code objects, %d symbols, no packed-fp32 op_sel instruction" % (path, nobj, nkern))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
