// lzo_device.h -- LZO1X chunks (compression.rs:174-183: lzokay_native::decompress_all), one wavefront per chunk.
//
// The format (Linux kernel, Documentation/staging/lzo.rst; lzokay's decompress() follows it): a first byte that may announce
// literals, then instructions -- M2 (1LLDDDSS / 01LDDDSS + H), M3 (001LLLLL [+ length bytes] + LE16), M4 (0001HLLL [+ length
// bytes] + LE16; distance 16384 ends the stream) and, by the count of literals the instruction before copied ("state"), a long
// literal run or a 2- / 3-byte match (0000xxxx) --, each followed by S = 0..3 literals.  The chain of instructions is serial
// (an instruction's meaning depends on the state the one before left) and ORC's LZO is rare (Java writers of a decade ago):
// this decoder parses wave-uniformly out of an LDS stage of the input and lets the 64 lanes do the copying through the LDS
// ring of decompress_kernels.hip -- correct for every stream the crate accepts, rejecting every stream it rejects (input /
// output overrun, look-behind before the output, a stream that does not end in the M4 marker, input left over), not fast.
#pragma once

__device__ __forceinline__ int lzo_wave(const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t lane, uint32_t* out_len, LzLds L) {
  if (n < 3) return 1;
  LzIn in{src, n, L.stage, 0};
  lzin_stage(in, 0, lane);
  LzOut o{L.ring, L.rsize - 1, dst, 0, 0};
  uint32_t ip = 0, state = 0, nstate = 0, lblen = 0, lbdist = 0;
  auto byte_at = [&](uint32_t pos) -> uint32_t { return (uint32_t)(lzin_peek(in, pos, lane) & 0xffu); };
  // zero bytes at ip (the long forms of a length): returns their count, ip moves behind them
  auto zeros = [&]() -> uint32_t {
    const uint32_t z0 = ip;
    while (ip < n) {
      const uint64_t w = lzin_peek(in, ip, lane);  // (zero beyond n: the bound above ends the walk there)
      if (w == 0 && ip + 8 <= n) {
        ip += 8;
        continue;
      }
      uint32_t k = w ? (uint32_t)__builtin_ctzll(w) >> 3 : 8u;
      if (ip + k > n) k = n - ip;
      ip += k;
      break;
    }
    return ip - z0;
  };
#define LZO_IN(k) \
  if (n - ip < (uint32_t)(k)) return 2
#define LZO_OUT(k) \
  if ((uint64_t)cap - o.out < (uint64_t)(k)) return 3
  {
    const uint32_t b0 = byte_at(0);
    if (b0 >= 22) {
      const uint32_t len = b0 - 17;
      ip = 1;
      LZO_IN(len);
      LZO_OUT(len);
      lzin_literal(in, o, ip, len, lane);
      ip += len;
      state = 4;
    } else if (b0 >= 18) {
      nstate = b0 - 17;
      state = nstate;
      ip = 1;
      LZO_IN(nstate);
      LZO_OUT(nstate);
      lzin_literal(in, o, ip, nstate, lane);
      ip += nstate;
    }
  }
  for (;;) {
    LZO_IN(1);
    const uint64_t w = lzin_peek(in, ip, lane);
    const uint32_t inst = (uint32_t)(w & 0xff), b1 = (uint32_t)(w >> 8) & 0xff, b2 = (uint32_t)(w >> 16) & 0xff;
    ip++;
    if (inst & 0xC0) {
      LZO_IN(1);
      ip++;
      lbdist = (b1 << 3) + ((inst >> 2) & 7) + 1;
      lblen = (inst >> 5) + 1;
      nstate = inst & 3;
    } else if (inst & 0x30) {
      const bool m3 = inst & 0x20;
      lblen = (inst & (m3 ? 0x1fu : 7u)) + 2;
      uint32_t lo = b1, hi = b2;
      if (lblen == 2) {
        const uint32_t z = zeros();
        if (z > 0x00800000u) return 4;  // (a length no chunk holds)
        LZO_IN(1);
        const uint64_t w2 = lzin_peek(in, ip, lane);
        lblen += z * 255 + (m3 ? 31 : 7) + (uint32_t)(w2 & 0xff);
        ip++;
        lo = (uint32_t)(w2 >> 8) & 0xff;
        hi = (uint32_t)(w2 >> 16) & 0xff;
      }
      LZO_IN(2);
      ip += 2;
      const uint32_t le16 = lo | hi << 8;
      nstate = le16 & 3;
      if (m3) {
        lbdist = (le16 >> 2) + 1;
      } else {
        lbdist = ((inst & 8) << 11) + (le16 >> 2);
        if (lbdist == 0) break;  // the stream's end marker
        lbdist += 16384;
      }
    } else if (state == 0) {
      uint32_t len = inst + 3;
      if (len == 3) {
        const uint32_t z = zeros();
        if (z > 0x00800000u) return 4;
        LZO_IN(1);
        len += z * 255 + 15 + byte_at(ip);
        ip++;
      }
      LZO_IN(len);
      LZO_OUT(len);
      lzin_literal(in, o, ip, len, lane);
      ip += len;
      state = 4;
      continue;
    } else {
      LZO_IN(1);
      ip++;
      nstate = inst & 3;
      lbdist = (inst >> 2) + (b1 << 2) + (state != 4 ? 1u : 2049u);
      lblen = state != 4 ? 2 : 3;
    }
    if ((uint64_t)lbdist > o.out) return 5;  // look-behind before the start of the output
    LZO_IN(nstate);
    LZO_OUT(lblen + nstate);
    lz_match(o, lbdist, lblen, lane);
    state = nstate;
    if (nstate) {
      lzin_literal(in, o, ip, nstate, lane);
      ip += nstate;
    }
  }
#undef LZO_IN
#undef LZO_OUT
  if (lblen != 3) return 6;  // the terminating M4 has length 3
  if (ip != n) return 7;     // input left over
  lz_flush(o, lane);
  *out_len = (uint32_t)o.out;
  return 0;
}
