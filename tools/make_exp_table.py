"""Generates csrc/gpis_exp_table.inc: the 2^(k/128) table of glibc's double-precision exp (sysdeps/ieee754/dbl-64/e_exp_data.c:
2^(k/N) = H[k] (1 + T[k]) with H[k] = the nearest double and T[k] the nearest double of the remainder, stored as
{bits of T[k], bits of H[k] - (k << 45)}).  The values are determined by that definition alone; mpmath computes them to 400 bits.
csrc/gpis_libm.hpp's exp replica is checked bit for bit against the host's exp() in tests/test_libm_replica_cpu.py.
usage: python tools/make_exp_table.py > sparse-conv-gpis-tungsten_amd/csrc/gpis_exp_table.inc"""
import struct, mpmath as mp
mp.mp.prec = 400
def d2u(x): return struct.unpack('<Q', struct.pack('<d', x))[0]
rows = []
for k in range(128):
    v = mp.power(2, mp.mpf(k) / 128)
    H = float(v)
    T = float(v / mp.mpf(H) - 1)
    rows.append((d2u(T), (d2u(H) - (k << 45)) & 0xFFFFFFFFFFFFFFFF))
print("/* 2^(k/128) = H[k] (1 + T[k]): {bits of T[k], bits of H[k] - (k << 45)}, k = 0 .. 127 */")
for k in range(0, 128, 2):
    print("    " + " ".join("0x%016xull, 0x%016xull," % r for r in rows[k:k + 2]))
