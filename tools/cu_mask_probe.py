"""Which CU-mask bits belong to which XCD (hipExtStreamCreateWithCUMask on MI355X): launches a probe kernel on streams with
candidate masks and prints where its workgroups ran.   python tools/cu_mask_probe.py"""
import collections
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import _lib

lib = _lib.load()


def probe(bits, nblocks=64, lds=100 * 1024, spin=200):
    words = [0] * 8
    for b in bits or []:
        words[b // 32] |= 1 << (b % 32)
    m = (C.c_uint * 8)(*words)
    out = (C.c_int * nblocks)()
    rc = lib.mitdvp_cu_mask_probe(0, m if bits is not None else None, 8 if bits is not None else 0, nblocks, lds, spin, out)
    assert rc == 0, rc
    return list(out)


def show(tag, bits, **kw):
    o = probe(bits, **kw)
    x = collections.Counter(v & 0xF for v in o)
    cus = len(set(o))
    print(f"{tag}: XCC histogram {dict(sorted(x.items()))}, distinct (xcc, cu) {cus}")


show("no mask, 256 blocks", None, nblocks=256)
show("bits 0..31 (contiguous)", list(range(32)))
show("bits 32..63 (contiguous)", list(range(32, 64)))
show("bits r=0 mod 8 (interleaved)", list(range(0, 256, 8)))
show("bits r=1 mod 8 (interleaved)", list(range(1, 256, 8)))
show("bits 0..31, 64 blocks (two rounds?)", list(range(32)), nblocks=64)


def where(tag, bits, nblocks=16):
    o = probe(bits, nblocks=nblocks, spin=50)
    s = sorted(set((v & 0xF, v >> 8) for v in o))
    print(f"{tag}: {len(s)} CUs:", " ".join(f"x{x}:cu{c:02x}" for x, c in s))


for b in (0, 1, 2, 7, 8, 9, 16, 31, 32, 63, 64, 128, 255):
    where(f"bit {b}", [b])
where("bits 0-7", list(range(8)))
where("bits 0,8,16,24", [0, 8, 16, 24], nblocks=32)
