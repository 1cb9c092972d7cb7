// audio_io.c — container decode for the ASR hot path.
//
// Replaces what the reference gets from `librosa.load(audio_file, sr=16000, mono=True)`
// (reference main.py:579-581): FLAC or RIFF/WAVE bytes -> mono float32 PCM in [-1,1)
// (int sample / 2^(bps-1), channels averaged), plus the sample rate.  Resampling is NOT
// done here: the reference's fixtures and the Willow device stream are 16 kHz already;
// any other rate is reported to the caller, which resamples to 16 kHz on the host
// (wis_hip.audio.resample: polyphase Kaiser-windowed sinc).
//
// Host-only plain C; no dependency.  The FLAC side implements the full subframe set
// (CONSTANT / VERBATIM / FIXED 0-4 / LPC 1-32, Rice methods 0 and 1 with escape
// partitions, wasted bits, all four channel assignments, 4..32 bps) and verifies the
// frame CRC-16 and the STREAMINFO MD5 of the decoded signal, which is the
// known-answer test for the three reference clips.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "wis_hip.h"

// ----------------------------------------------------------------------------------------
// MD5 (RFC 1321) — only used to verify STREAMINFO's signature of the decoded PCM.
typedef struct { uint32_t a, b, c, d; uint64_t len; uint8_t buf[64]; uint32_t fill; } md5_t;
static const uint32_t MD5_K[64] = {
  0xd76aa478,0xe8c7b756,0x242070db,0xc1bdceee,0xf57c0faf,0x4787c62a,0xa8304613,0xfd469501,
  0x698098d8,0x8b44f7af,0xffff5bb1,0x895cd7be,0x6b901122,0xfd987193,0xa679438e,0x49b40821,
  0xf61e2562,0xc040b340,0x265e5a51,0xe9b6c7aa,0xd62f105d,0x02441453,0xd8a1e681,0xe7d3fbc8,
  0x21e1cde6,0xc33707d6,0xf4d50d87,0x455a14ed,0xa9e3e905,0xfcefa3f8,0x676f02d9,0x8d2a4c8a,
  0xfffa3942,0x8771f681,0x6d9d6122,0xfde5380c,0xa4beea44,0x4bdecfa9,0xf6bb4b60,0xbebfbc70,
  0x289b7ec6,0xeaa127fa,0xd4ef3085,0x04881d05,0xd9d4d039,0xe6db99e5,0x1fa27cf8,0xc4ac5665,
  0xf4292244,0x432aff97,0xab9423a7,0xfc93a039,0x655b59c3,0x8f0ccc92,0xffeff47d,0x85845dd1,
  0x6fa87e4f,0xfe2ce6e0,0xa3014314,0x4e0811a1,0xf7537e82,0xbd3af235,0x2ad7d2bb,0xeb86d391};
static const uint8_t MD5_S[64] = {7,12,17,22,7,12,17,22,7,12,17,22,7,12,17,22,
  5,9,14,20,5,9,14,20,5,9,14,20,5,9,14,20,4,11,16,23,4,11,16,23,4,11,16,23,4,11,16,23,
  6,10,15,21,6,10,15,21,6,10,15,21,6,10,15,21};
static void md5_block(md5_t* m, const uint8_t* p) {
  uint32_t w[16];
  for (int i = 0; i < 16; ++i)
    w[i] = (uint32_t)p[4*i] | ((uint32_t)p[4*i+1] << 8) | ((uint32_t)p[4*i+2] << 16) | ((uint32_t)p[4*i+3] << 24);
  uint32_t a = m->a, b = m->b, c = m->c, d = m->d;
  for (int i = 0; i < 64; ++i) {
    uint32_t f; int g;
    if (i < 16)      { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5*i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d;          g = (3*i + 5) & 15; }
    else             { f = c ^ (b | ~d);       g = (7*i) & 15; }
    uint32_t t = a + f + MD5_K[i] + w[g];
    a = d; d = c; c = b;
    b = b + ((t << MD5_S[i]) | (t >> (32 - MD5_S[i])));
  }
  m->a += a; m->b += b; m->c += c; m->d += d;
}
static void md5_init(md5_t* m) { m->a = 0x67452301; m->b = 0xefcdab89; m->c = 0x98badcfe; m->d = 0x10325476; m->len = 0; m->fill = 0; }
static void md5_update(md5_t* m, const uint8_t* p, size_t n) {
  m->len += n;
  while (n) {
    size_t k = 64 - m->fill; if (k > n) k = n;
    memcpy(m->buf + m->fill, p, k); m->fill += (uint32_t)k; p += k; n -= k;
    if (m->fill == 64) { md5_block(m, m->buf); m->fill = 0; }
  }
}
static void md5_final(md5_t* m, uint8_t out[16]) {
  uint64_t bits = m->len * 8; uint8_t pad = 0x80;
  md5_update(m, &pad, 1); pad = 0;
  while (m->fill != 56) md5_update(m, &pad, 1);
  uint8_t l[8]; for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8*i));
  md5_update(m, l, 8);
  uint32_t v[4] = {m->a, m->b, m->c, m->d};
  for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i/4] >> (8*(i%4)));
}

// ----------------------------------------------------------------------------------------
// bit reader (MSB first)
typedef struct { const uint8_t* p; size_t n; size_t pos; /* bit position */ int err; } br_t;
// the next 57+ bits of the stream, left-aligned in a 64-bit word (bits past the end of the buffer read as zero; br_bits / br_unary
// flag the overrun): one unaligned 8-byte load and a byte swap instead of a loop over single bits - the container decode is the
// largest host-side cost of a short request (tools/host_ceiling.py)
static inline uint64_t br_peek(const br_t* b) {
  const size_t byte = b->pos >> 3;
  uint64_t w;
  if (byte + 8 <= b->n) { memcpy(&w, b->p + byte, 8); w = __builtin_bswap64(w); }
  else { w = 0; for (size_t i = 0; i < 8; ++i) w = (w << 8) | (byte + i < b->n ? b->p[byte + i] : 0u); }
  return w << (b->pos & 7);
}
static inline uint32_t br_bits(br_t* b, int n) {  // n in 0..32
  if (n == 0) return 0;
  if (b->pos + (size_t)n > b->n * 8) { b->err = 1; b->pos = b->n * 8; return 0; }
  const uint32_t v = (uint32_t)(br_peek(b) >> (64 - n));
  b->pos += (size_t)n;
  return v;
}
static inline int32_t br_sbits(br_t* b, int n) {
  if (n == 0) return 0;
  uint32_t v = br_bits(b, n);
  if (n < 32 && (v >> (n - 1))) v |= ~0u << n;
  return (int32_t)v;
}
static inline uint32_t br_unary(br_t* b) {  // count zeros before a one
  uint32_t q = 0;
  for (;;) {
    if (b->pos >= b->n * 8) { b->err = 1; return 0; }
    const uint64_t w = br_peek(b) | ((uint64_t)1 << 7);      // 57 valid bits per peek: a stop bit below them bounds the count
    const int z = __builtin_clzll(w);
    if (z < 56) {
      if (b->pos + (size_t)z + 1 > b->n * 8) { b->err = 1; b->pos = b->n * 8; return 0; }
      b->pos += (size_t)z + 1;
      return q + (uint32_t)z;
    }
    q += 56; b->pos += 56;
  }
}
// CRC-8 (poly 0x07) and CRC-16 (poly 0x8005) of the frame headers / frames, byte-wise by table
static uint8_t CRC8_T[256];
static uint16_t CRC16_T[256];
__attribute__((constructor)) static void crc_tables(void) {
  for (int i = 0; i < 256; ++i) {
    uint8_t c = (uint8_t)i; for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
    CRC8_T[i] = c;
    uint16_t d = (uint16_t)(i << 8); for (int k = 0; k < 8; ++k) d = (uint16_t)((d & 0x8000) ? (d << 1) ^ 0x8005 : (d << 1));
    CRC16_T[i] = d;
  }
}
static uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) c = CRC8_T[c ^ p[i]];
  return c;
}
static uint16_t crc16(const uint8_t* p, size_t n) {
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ CRC16_T[(c >> 8) ^ p[i]]);
  return c;
}

static int flac_residual(br_t* b, int64_t* out, int blocksize, int order) {
  int method = (int)br_bits(b, 2);
  if (method > 1) return -1;
  int pbits = method ? 5 : 4, esc = method ? 31 : 15;
  int porder = (int)br_bits(b, 4);
  int nparts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder) return -1;
  int idx = order;
  for (int p = 0; p < nparts; ++p) {
    int cnt = (porder == 0) ? blocksize - order : (p == 0 ? (blocksize >> porder) - order : (blocksize >> porder));
    if (cnt < 0) return -1;
    int k = (int)br_bits(b, pbits);
    if (k == esc) {
      int nb = (int)br_bits(b, 5);
      for (int i = 0; i < cnt; ++i) out[idx++] = br_sbits(b, nb);
    } else {
      for (int i = 0; i < cnt; ++i) {
        uint32_t q = br_unary(b);
        uint32_t r = k ? br_bits(b, k) : 0;
        uint32_t u = (q << k) | r;
        out[idx++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (b->err) return -1;
  }
  return 0;
}

static int flac_subframe(br_t* b, int64_t* out, int blocksize, int bps) {
  if (br_bits(b, 1)) return -1;  // padding bit must be 0
  int type = (int)br_bits(b, 6);
  int wasted = 0;
  if (br_bits(b, 1)) { wasted = 1 + (int)br_unary(b); bps -= wasted; }
  if (bps <= 0) return -1;
  if (type == 0) {  // CONSTANT
    int64_t v = br_sbits(b, bps > 32 ? 32 : bps);
    for (int i = 0; i < blocksize; ++i) out[i] = v;
  } else if (type == 1) {  // VERBATIM
    for (int i = 0; i < blocksize; ++i) out[i] = br_sbits(b, bps > 32 ? 32 : bps);
  } else if (type >= 8 && type <= 12) {  // FIXED
    int order = type - 8;
    if (order > blocksize) return -1;
    for (int i = 0; i < order; ++i) out[i] = br_sbits(b, bps > 32 ? 32 : bps);
    if (flac_residual(b, out, blocksize, order)) return -1;
    for (int i = order; i < blocksize; ++i) {
      int64_t p = 0;
      switch (order) {
        case 1: p = out[i-1]; break;
        case 2: p = 2*out[i-1] - out[i-2]; break;
        case 3: p = 3*out[i-1] - 3*out[i-2] + out[i-3]; break;
        case 4: p = 4*out[i-1] - 6*out[i-2] + 4*out[i-3] - out[i-4]; break;
        default: break;
      }
      out[i] += p;
    }
  } else if (type >= 32) {  // LPC
    int order = (type & 31) + 1;
    if (order > blocksize) return -1;
    for (int i = 0; i < order; ++i) out[i] = br_sbits(b, bps > 32 ? 32 : bps);
    int prec = (int)br_bits(b, 4) + 1;
    if (prec == 16) return -1;
    int shift = br_sbits(b, 5);
    if (shift < 0) return -1;
    int32_t coef[32];
    for (int i = 0; i < order; ++i) coef[i] = br_sbits(b, prec);
    if (flac_residual(b, out, blocksize, order)) return -1;
    for (int i = order; i < blocksize; ++i) {
      int64_t s = 0;
      for (int j = 0; j < order; ++j) s += (int64_t)coef[j] * out[i-1-j];
      out[i] += s >> shift;
    }
  } else {
    return -1;  // reserved
  }
  if (wasted) for (int i = 0; i < blocksize; ++i) out[i] *= ((int64_t)1 << wasted);
  return b->err ? -1 : 0;
}

// Decodes a whole FLAC stream.  Returns 0 or a negative WIS_E_* code.
static int flac_decode(const uint8_t* data, size_t n, float** pcm_out, int64_t* n_out, int* sr_out, int* md5_ok) {
  if (n < 42 || memcmp(data, "fLaC", 4)) return WIS_E_FORMAT;
  size_t off = 4; int last = 0;
  int have_si = 0; int sr = 0, ch = 0, bps = 0; uint64_t total = 0; uint8_t md5sig[16] = {0};
  int max_bs = 0;
  while (!last) {
    if (off + 4 > n) return WIS_E_FORMAT;
    last = data[off] >> 7; int btype = data[off] & 0x7f;
    size_t blen = ((size_t)data[off+1] << 16) | ((size_t)data[off+2] << 8) | data[off+3];
    off += 4;
    if (off + blen > n) return WIS_E_FORMAT;
    if (btype == 0) {
      if (blen < 34) return WIS_E_FORMAT;
      const uint8_t* s = data + off;
      max_bs = (s[2] << 8) | s[3];
      sr = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      ch = ((s[12] >> 1) & 7) + 1;
      bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      total = ((uint64_t)(s[13] & 15) << 32) | ((uint64_t)s[14] << 24) | ((uint64_t)s[15] << 16) | ((uint64_t)s[16] << 8) | s[17];
      memcpy(md5sig, s + 18, 16);
      have_si = 1;
    }
    off += blen;
  }
  if (!have_si || ch < 1 || ch > 8 || bps < 4 || bps > 32 || max_bs < 16) return WIS_E_FORMAT;

  size_t cap = total ? (size_t)total : (size_t)1 << 20;
  float* pcm = (float*)malloc(cap * sizeof(float) + 16);
  int64_t* sub = (int64_t*)malloc((size_t)ch * 65536 * sizeof(int64_t));
  if (!pcm || !sub) { free(pcm); free(sub); return WIS_E_NOMEM; }
  md5_t md5; md5_init(&md5);
  size_t nsamp = 0; int rc = 0;
  const float scale = 1.0f / (float)((int64_t)1 << (bps - 1));
  const int bytes_ps = (bps + 7) / 8;

  while (off + 2 <= n) {
    if (!(data[off] == 0xff && (data[off+1] & 0xfe) == 0xf8)) { rc = WIS_E_FORMAT; break; }
    br_t b = {data + off, n - off, 0, 0};
    br_bits(&b, 15); br_bits(&b, 1);
    int bs_code = (int)br_bits(&b, 4), sr_code = (int)br_bits(&b, 4);
    int ch_code = (int)br_bits(&b, 4), bps_code = (int)br_bits(&b, 3);
    br_bits(&b, 1);
    // UTF-8 style coded frame/sample number
    uint32_t lead = br_bits(&b, 8); int extra = 0;
    if (lead & 0x80) { while (lead & (0x80u >> extra)) ++extra; if (extra < 2 || extra > 7) { rc = WIS_E_FORMAT; break; } --extra; }
    for (int i = 0; i < extra; ++i) br_bits(&b, 8);
    int blocksize = 0;
    if (bs_code == 1) blocksize = 192;
    else if (bs_code >= 2 && bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br_bits(&b, 8) + 1;
    else if (bs_code == 7) blocksize = (int)br_bits(&b, 16) + 1;
    else if (bs_code >= 8) blocksize = 256 << (bs_code - 8);
    else { rc = WIS_E_FORMAT; break; }
    if (sr_code == 12) br_bits(&b, 8); else if (sr_code == 13 || sr_code == 14) br_bits(&b, 16);
    static const int bps_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int fbps = bps_tab[bps_code]; if (fbps == 0) fbps = bps; if (fbps < 0) { rc = WIS_E_FORMAT; break; }
    size_t hdr_bytes = b.pos >> 3;
    uint32_t c8 = br_bits(&b, 8);
    if (b.err || crc8(data + off, hdr_bytes) != c8) { rc = WIS_E_FORMAT; break; }
    int nch = (ch_code < 8) ? ch_code + 1 : 2;
    if (ch_code > 10 || nch != ch || blocksize > 65535) { rc = WIS_E_FORMAT; break; }
    for (int c = 0; c < nch && !rc; ++c) {
      int sbps = fbps;
      if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) sbps += 1;  // side channel
      if (flac_subframe(&b, sub + (size_t)c * 65536, blocksize, sbps)) rc = WIS_E_FORMAT;
    }
    if (rc) break;
    b.pos = (b.pos + 7) & ~(size_t)7;
    size_t body = b.pos >> 3;
    uint32_t c16 = br_bits(&b, 16);
    if (b.err || crc16(data + off, body) != c16) { rc = WIS_E_FORMAT; break; }
    off += body + 2;
    int64_t* s0 = sub; int64_t* s1 = sub + 65536;
    if (ch_code == 8)       for (int i = 0; i < blocksize; ++i) s1[i] = s0[i] - s1[i];
    else if (ch_code == 9)  for (int i = 0; i < blocksize; ++i) s0[i] = s0[i] + s1[i];
    else if (ch_code == 10) for (int i = 0; i < blocksize; ++i) {
      int64_t mid = s0[i], side = s1[i];
      mid = (mid << 1) | (side & 1);
      s0[i] = (mid + side) >> 1; s1[i] = (mid - side) >> 1;
    }
    if (nsamp + (size_t)blocksize > cap) {
      cap = (nsamp + blocksize) * 2;
      float* np_ = (float*)realloc(pcm, cap * sizeof(float) + 16);
      if (!np_) { rc = WIS_E_NOMEM; break; }
      pcm = np_;
    }
    for (int i = 0; i < blocksize; ++i) {
      // mono mix = mean over channels in float, matching librosa's to_mono on /2^(bps-1) floats
      float acc = 0.f;
      for (int c = 0; c < nch; ++c) {
        int64_t v = sub[(size_t)c * 65536 + i];
        uint8_t le[4]; for (int k = 0; k < bytes_ps; ++k) le[k] = (uint8_t)((uint64_t)v >> (8*k));
        md5_update(&md5, le, (size_t)bytes_ps);
        acc += (float)v * scale;
      }
      pcm[nsamp + i] = nch == 1 ? acc : acc / (float)nch;
    }
    nsamp += (size_t)blocksize;
  }
  free(sub);
  if (rc) { free(pcm); return rc; }
  if (total && nsamp != total) { free(pcm); return WIS_E_FORMAT; }
  uint8_t dig[16]; md5_final(&md5, dig);
  int zero = 1; for (int i = 0; i < 16; ++i) if (md5sig[i]) zero = 0;
  *md5_ok = zero ? -1 : (memcmp(dig, md5sig, 16) == 0);
  if (*md5_ok == 0) { free(pcm); return WIS_E_CHECKSUM; }
  *pcm_out = pcm; *n_out = (int64_t)nsamp; *sr_out = sr;
  return 0;
}

// ----------------------------------------------------------------------------------------
// RIFF/WAVE: PCM 8/16/24/32-bit and IEEE float32, any channel count (averaged to mono).
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static int wav_decode(const uint8_t* d, size_t n, float** pcm_out, int64_t* n_out, int* sr_out) {
  if (n < 12 || memcmp(d, "RIFF", 4) || memcmp(d + 8, "WAVE", 4)) return WIS_E_FORMAT;
  size_t off = 12; int fmt = 0, ch = 0, sr = 0, bits = 0; const uint8_t* data = NULL; size_t dlen = 0;
  while (off + 8 <= n) {
    uint32_t len = rd32(d + off + 4); const uint8_t* body = d + off + 8;
    size_t avail = n - off - 8; if (len > avail) len = (uint32_t)avail;  // streaming writers leave 0xffffffff
    if (!memcmp(d + off, "fmt ", 4) && len >= 16) {
      fmt = body[0] | (body[1] << 8); ch = body[2] | (body[3] << 8); sr = (int)rd32(body + 4); bits = body[14] | (body[15] << 8);
      if (fmt == 0xfffe && len >= 26) fmt = body[24] | (body[25] << 8);  // WAVE_FORMAT_EXTENSIBLE sub-format
    } else if (!memcmp(d + off, "data", 4)) { data = body; dlen = len; break; }
    off += 8 + (size_t)len + (len & 1);
  }
  if (!data || ch < 1 || !(fmt == 1 || fmt == 3)) return WIS_E_FORMAT;
  if (fmt == 3 && bits != 32) return WIS_E_FORMAT;
  if (fmt == 1 && bits != 8 && bits != 16 && bits != 24 && bits != 32) return WIS_E_FORMAT;
  int bps = bits / 8; size_t frames = dlen / ((size_t)bps * ch);
  float* pcm = (float*)malloc(frames * sizeof(float) + 16);
  if (!pcm) return WIS_E_NOMEM;
  for (size_t i = 0; i < frames; ++i) {
    float acc = 0.f;
    for (int c = 0; c < ch; ++c) {
      const uint8_t* p = data + (i * ch + c) * bps; float v;
      if (fmt == 3) { uint32_t u = rd32(p); memcpy(&v, &u, 4); }
      else if (bits == 8)  v = ((int)p[0] - 128) / 128.0f;
      else if (bits == 16) v = (int16_t)(p[0] | (p[1] << 8)) / 32768.0f;
      else if (bits == 24) v = (float)(((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24)) >> 8) / 8388608.0f;
      else v = (float)((int32_t)rd32(p)) / 2147483648.0f;
      acc += v;
    }
    pcm[i] = ch == 1 ? acc : acc / (float)ch;
  }
  *pcm_out = pcm; *n_out = (int64_t)frames; *sr_out = sr;
  return 0;
}

// ----------------------------------------------------------------------------------------
// C-ABI (declared in include/wis_hip.h)
int wis_audio_decode(const void* bytes, size_t n_bytes, float** pcm_out, int64_t* n_samples_out,
                     int* sample_rate_out, int* md5_status_out) {
  if (!bytes || !pcm_out || !n_samples_out || !sample_rate_out) return WIS_E_ARG;
  const uint8_t* d = (const uint8_t*)bytes; int md5 = -1; int rc;
  if (n_bytes >= 4 && !memcmp(d, "fLaC", 4)) rc = flac_decode(d, n_bytes, pcm_out, n_samples_out, sample_rate_out, &md5);
  else if (n_bytes >= 4 && !memcmp(d, "RIFF", 4)) rc = wav_decode(d, n_bytes, pcm_out, n_samples_out, sample_rate_out);
  else rc = WIS_E_FORMAT;
  if (md5_status_out) *md5_status_out = md5;
  return rc;
}
void wis_audio_free(float* pcm) { free(pcm); }
