// Encoded page -> RGB8.  See image_decode.h for what this replaces in the reference
// (retto-core/src/image_helper.rs:34-44) and which conversion rules it follows.
#include "image_decode.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/retto_hip.h"
#include "common.h"

namespace rt {
namespace {

[[noreturn]] void bad(const std::string& why) { throw RtError(RT_ERR_IMAGE, "image decode: " + why); }

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t be16(const uint8_t* p) { return ((uint32_t)p[0] << 8) | p[1]; }
inline uint32_t le32(const uint8_t* p) { return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0]; }
inline uint32_t le16(const uint8_t* p) { return ((uint32_t)p[1] << 8) | p[0]; }
inline uint8_t u16_to_u8(uint32_t v) { return (uint8_t)((v + 128) / 257); }  // image 0.25.6: FromPrimitive<u16> for u8

constexpr size_t MAX_PIXELS = (size_t)1 << 28;  // 268 M pixels: far above any page, stops absurd headers before allocating

void check_dims(uint64_t w, uint64_t h) {
  if (w == 0 || h == 0) bad("zero image dimension");
  if (w > 0x7fffffffu || h > 0x7fffffffu || w * h > MAX_PIXELS) bad("image dimensions too large");
}

// ------------------------------------------------------------------------------------------------
// PNG (ISO/IEC 15948): all colour types and bit depths, Adam7 interlace, CRC checked
// ------------------------------------------------------------------------------------------------
void decode_png(const uint8_t* d, size_t n, std::vector<uint8_t>* rgb, int* oh, int* ow) {
  size_t pos = 8;
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool seen_end = false;
  while (!seen_end) {
    if (pos + 12 > n) bad("PNG: truncated chunk stream");
    const uint32_t len = be32(d + pos);
    const uint8_t* type = d + pos + 4;
    if ((size_t)len > n - pos - 12) bad("PNG: chunk length past the end of the data");
    const uint8_t* body = d + pos + 8;
    const uint32_t crc = be32(body + len);
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, len + 4) != crc) bad("PNG: CRC mismatch in chunk " + std::string((const char*)type, 4));
    if (ctype < 0 && memcmp(type, "IHDR", 4) != 0) bad("PNG: first chunk is not IHDR");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13 || ctype >= 0) bad("PNG: bad IHDR");
      W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      check_dims(W, H);
      static const int ok_depths[7] = {0x1f /*1,2,4,8,16*/, 0, 0x18 /*8,16*/, 0x0f /*1,2,4,8*/, 0x18, 0, 0x18};
      int bit = depth == 1 ? 1 : depth == 2 ? 2 : depth == 4 ? 4 : depth == 8 ? 8 : depth == 16 ? 16 : 0;
      if (ctype > 6 || !bit || !(ok_depths[ctype] & bit)) bad("PNG: invalid colour type / bit depth combination");
      if (body[10] != 0 || body[11] != 0 || interlace > 1) bad("PNG: unknown compression, filter or interlace method");
    } else if (!memcmp(type, "PLTE", 4)) {
      if (len % 3 != 0 || len > 768) bad("PNG: bad PLTE length");
      plte.assign(body, body + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!memcmp(type, "IEND", 4)) {
      seen_end = true;
    } else if (!(type[0] & 0x20)) {
      bad("PNG: unknown critical chunk " + std::string((const char*)type, 4));
    }
    pos += 12 + (size_t)len;
  }
  if (ctype < 0 || idat.empty()) bad("PNG: no image data");
  if (ctype == 3 && plte.empty()) bad("PNG: palette image without PLTE");
  const int chans = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
  const int bpp_bits = chans * depth, bpp = std::max(1, bpp_bits / 8);
  static const int ax0[7] = {0, 4, 0, 2, 0, 1, 0}, ay0[7] = {0, 0, 4, 0, 2, 0, 1};
  static const int adx[7] = {8, 8, 4, 4, 2, 2, 1}, ady[7] = {8, 8, 8, 4, 4, 2, 2};
  const int npass = interlace ? 7 : 1;
  size_t raw_size = 0;
  for (int p = 0; p < npass; p++) {
    const size_t pw = interlace ? ((size_t)W - ax0[p] + adx[p] - 1) / adx[p] : W;
    const size_t ph = interlace ? ((size_t)H - ay0[p] + ady[p] - 1) / ady[p] : H;
    if ((interlace && ((size_t)W <= (size_t)ax0[p] || (size_t)H <= (size_t)ay0[p])) || pw == 0 || ph == 0) continue;
    raw_size += ph * (1 + (pw * bpp_bits + 7) / 8);
  }
  std::vector<uint8_t> raw(raw_size);
  {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) bad("PNG: zlib init failed");
    zs.next_in = idat.data();
    zs.next_out = raw.data();
    size_t in_left = idat.size(), out_left = raw.size();
    int rc = Z_OK;
    while (rc == Z_OK && out_left > 0) {
      zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
      zs.avail_out = (uInt)std::min<size_t>(out_left, 1u << 30);
      const uInt in0 = zs.avail_in, out0 = zs.avail_out;
      rc = inflate(&zs, Z_NO_FLUSH);
      in_left -= in0 - zs.avail_in;
      out_left -= out0 - zs.avail_out;
      if (rc == Z_BUF_ERROR || (rc == Z_OK && in0 == zs.avail_in && out0 == zs.avail_out)) break;
    }
    inflateEnd(&zs);
    if (rc != Z_OK && rc != Z_STREAM_END) bad("PNG: corrupt deflate stream");
    if (out_left != 0) bad("PNG: image data ends early");
  }
  rgb->assign((size_t)W * H * 3, 0);
  const uint32_t gray_mul = depth < 8 ? 255u / ((1u << depth) - 1u) : 1u;
  const size_t npal = plte.size() / 3;
  size_t off = 0;
  std::vector<uint8_t> prev, cur;
  for (int p = 0; p < npass; p++) {
    if (interlace && ((size_t)W <= (size_t)ax0[p] || (size_t)H <= (size_t)ay0[p])) continue;
    const size_t pw = interlace ? ((size_t)W - ax0[p] + adx[p] - 1) / adx[p] : W;
    const size_t ph = interlace ? ((size_t)H - ay0[p] + ady[p] - 1) / ady[p] : H;
    if (pw == 0 || ph == 0) continue;
    const size_t rb = (pw * bpp_bits + 7) / 8;
    prev.assign(rb, 0);
    cur.resize(rb);
    for (size_t y = 0; y < ph; y++) {
      const int ft = raw[off];
      const uint8_t* src = raw.data() + off + 1;
      off += 1 + rb;
      switch (ft) {
        case 0: memcpy(cur.data(), src, rb); break;
        case 1:
          for (size_t i = 0; i < rb; i++) cur[i] = (uint8_t)(src[i] + (i >= (size_t)bpp ? cur[i - bpp] : 0));
          break;
        case 2:
          for (size_t i = 0; i < rb; i++) cur[i] = (uint8_t)(src[i] + prev[i]);
          break;
        case 3:
          for (size_t i = 0; i < rb; i++) cur[i] = (uint8_t)(src[i] + (((i >= (size_t)bpp ? cur[i - bpp] : 0) + prev[i]) >> 1));
          break;
        case 4:
          for (size_t i = 0; i < rb; i++) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
            const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
            cur[i] = (uint8_t)(src[i] + (pa <= pb && pa <= pc ? a : pb <= pc ? b : c));
          }
          break;
        default: bad("PNG: unknown filter type");
      }
      const size_t oy = interlace ? (size_t)ay0[p] + y * ady[p] : y;
      uint8_t* orow = rgb->data() + oy * W * 3;
      for (size_t x = 0; x < pw; x++) {
        const size_t ox = interlace ? (size_t)ax0[p] + x * adx[p] : x;
        uint8_t* o = orow + ox * 3;
        if (depth == 8) {
          const uint8_t* s = cur.data() + x * chans;
          if (ctype == 2 || ctype == 6) { o[0] = s[0]; o[1] = s[1]; o[2] = s[2]; }
          else if (ctype == 3) { if (s[0] < npal) { o[0] = plte[3 * s[0]]; o[1] = plte[3 * s[0] + 1]; o[2] = plte[3 * s[0] + 2]; } }
          else o[0] = o[1] = o[2] = s[0];
        } else if (depth == 16) {
          const uint8_t* s = cur.data() + x * chans * 2;
          if (ctype == 2 || ctype == 6) { o[0] = u16_to_u8(be16(s)); o[1] = u16_to_u8(be16(s + 2)); o[2] = u16_to_u8(be16(s + 4)); }
          else o[0] = o[1] = o[2] = u16_to_u8(be16(s));
        } else {  // 1, 2, 4 bits: grey or palette index, most significant bits first
          const size_t bit = x * depth;
          const uint32_t v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
          if (ctype == 3) { if (v < npal) { o[0] = plte[3 * v]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v + 2]; } }
          else o[0] = o[1] = o[2] = (uint8_t)(v * gray_mul);
        }
      }
      prev.swap(cur);
    }
  }
  *oh = (int)H; *ow = (int)W;
}

// ------------------------------------------------------------------------------------------------
// PNM: P2 / P3 (ASCII) and P5 / P6 (binary) with maxval 255 or 65535
// ------------------------------------------------------------------------------------------------
void decode_pnm(const uint8_t* d, size_t n, std::vector<uint8_t>* rgb, int* oh, int* ow) {
  const int kind = d[1] - '0';
  size_t pos = 2;
  auto token = [&]() -> long {
    for (;;) {
      while (pos < n && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t' || d[pos] == '\v' || d[pos] == '\f')) pos++;
      if (pos < n && d[pos] == '#') { while (pos < n && d[pos] != '\n') pos++; continue; }
      break;
    }
    if (pos >= n || d[pos] < '0' || d[pos] > '9') bad("PNM: malformed header or sample");
    long v = 0;
    while (pos < n && d[pos] >= '0' && d[pos] <= '9') { v = v * 10 + (d[pos++] - '0'); if (v > 0x7fffffffL) bad("PNM: number out of range"); }
    return v;
  };
  const long W = token(), H = token(), maxv = token();
  check_dims((uint64_t)W, (uint64_t)H);
  if (maxv != 255 && maxv != 65535) bad("PNM: only maxval 255 and 65535 are supported");
  const int chans = (kind == 3 || kind == 6) ? 3 : 1;
  const size_t count = (size_t)W * H * chans;
  rgb->assign((size_t)W * H * 3, 0);
  auto put = [&](size_t i, uint32_t v) {
    const uint8_t b = maxv == 255 ? (uint8_t)v : u16_to_u8(v);
    if (chans == 3) (*rgb)[i] = b;
    else { (*rgb)[3 * i] = (*rgb)[3 * i + 1] = (*rgb)[3 * i + 2] = b; }
  };
  if (kind == 2 || kind == 3) {
    for (size_t i = 0; i < count; i++) { const long v = token(); if (v > maxv) bad("PNM: sample above maxval"); put(i, (uint32_t)v); }
  } else {
    pos++;  // the single whitespace byte after maxval
    const size_t bytes = count * (maxv == 255 ? 1 : 2);
    if (pos > n || n - pos < bytes) bad("PNM: pixel data ends early");
    for (size_t i = 0; i < count; i++) put(i, maxv == 255 ? d[pos + i] : be16(d + pos + 2 * i));
  }
  *oh = (int)H; *ow = (int)W;
}

// ------------------------------------------------------------------------------------------------
// BMP: BITMAPINFOHEADER-family headers, uncompressed 8-bit palette / 24 / 32 bits per pixel
// ------------------------------------------------------------------------------------------------
void decode_bmp(const uint8_t* d, size_t n, std::vector<uint8_t>* rgb, int* oh, int* ow) {
  if (n < 54) bad("BMP: truncated header");
  const uint32_t data_off = le32(d + 10), hdr = le32(d + 14);
  if (hdr < 40 || 14 + (size_t)hdr > n) bad("BMP: unsupported header");
  const int32_t W = (int32_t)le32(d + 18), Hs = (int32_t)le32(d + 22);
  const int bits = (int)le16(d + 28);
  const uint32_t comp = le32(d + 30);
  const bool top_down = Hs < 0;
  const int64_t H = top_down ? -(int64_t)Hs : Hs;
  if (W <= 0) bad("BMP: bad width");
  check_dims((uint64_t)W, (uint64_t)H);
  if (!(comp == 0 || (comp == 3 && bits == 32)) || !(bits == 8 || bits == 24 || bits == 32)) bad("BMP: only uncompressed 8 / 24 / 32-bit images are supported");
  const size_t stride = (((size_t)W * bits + 31) / 32) * 4;
  if (data_off > n || (n - data_off) / stride < (size_t)H) bad("BMP: pixel data ends early");
  const uint8_t* pal = d + 14 + hdr;
  uint32_t ncol = le32(d + 46);
  if (bits == 8) {
    if (ncol == 0 || ncol > 256) ncol = 256;
    if (14 + (size_t)hdr + 4 * (size_t)ncol > n) bad("BMP: palette ends early");
  }
  rgb->assign((size_t)W * H * 3, 0);
  for (int64_t y = 0; y < H; y++) {
    const uint8_t* s = d + data_off + (size_t)(top_down ? y : H - 1 - y) * stride;
    uint8_t* o = rgb->data() + (size_t)y * W * 3;
    for (int32_t x = 0; x < W; x++, o += 3) {
      if (bits == 8) { const uint32_t i = s[x]; if (i < ncol) { o[0] = pal[4 * i + 2]; o[1] = pal[4 * i + 1]; o[2] = pal[4 * i]; } }
      else { const uint8_t* p = s + (size_t)x * (bits / 8); o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
    }
  }
  *oh = (int)H; *ow = (int)W;
}

// ------------------------------------------------------------------------------------------------
// JPEG (ITU-T T.81): Huffman DCT, sequential (SOF0 / SOF1) and progressive (SOF2), 8-bit, grey or three components, any
// sampling factors, interleaved or per-component scans, restart intervals.  Arithmetic follows
// the IJG conventions every mainstream decoder reproduces: the 13-bit fixed-point "slow integer" inverse
// DCT of Loeffler-Ligtenberg-Moschytz, triangle-filter ("fancy") chroma upsampling for 2:1 factors,
// 16-bit fixed-point YCbCr -> RGB.  Progressive files (SOF2: spectral selection + successive approximation)
// accumulate their coefficients over the scans and are transformed at the end.  Lossless, hierarchical,
// arithmetic-coded and CMYK files are rejected with a message.
// ------------------------------------------------------------------------------------------------
struct Huff {
  bool ok = false;
  uint8_t vals[256];
  uint16_t look[512];     // 9-bit lookahead: (length << 8) | symbol, 0 = longer code
  int32_t maxcode[18];    // largest code of each length (-1: none), [17] sentinel
  int32_t valoff[17];     // vals index of the first code of a length, minus that code
};
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct JComp {
  int id = 0, hs = 1, vs = 1, tq = 0, td = 0, ta = 0;
  int cw = 0, ch = 0;      // real sample dimensions: ceil(W * hs / hmax), ceil(H * vs / vmax)
  int stride = 0, rows = 0;  // allocated plane (whole MCUs)
  int pred = 0;
  std::vector<uint8_t> plane;
  std::vector<int> coef;   // progressive: all coefficients of the (MCU-padded) component, 64 per block, natural order
};

struct JpegDec {
  const uint8_t* d; size_t n, pos = 2;
  int W = 0, H = 0, nc = 0, hmax = 1, vmax = 1, restart = 0;
  bool have_sof = false, adobe = false; int adobe_tf = -1;
  JComp c[3];
  uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
  Huff dc[4], ac[4];
  uint64_t bits = 0; int nbits = 0; bool hit_marker = false;
  bool progressive = false; int Ss = 0, Se = 63, Ah = 0, Al = 0, eobrun = 0;

  void fill() {
    while (nbits <= 56) {
      uint32_t b = 0;
      if (!hit_marker && pos < n) {
        b = d[pos];
        if (b == 0xff) {
          if (pos + 1 < n && d[pos + 1] == 0x00) pos += 2;
          else { hit_marker = true; b = 0; }  // a marker ends the entropy-coded segment: feed zeros
        } else pos++;
      } else hit_marker = true;
      bits |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  inline int getbits(int s) {
    if (nbits < s) fill();
    const int v = (int)(bits >> (64 - s));
    bits <<= s; nbits -= s;
    return v;
  }
  inline int decode(const Huff& h) {
    if (nbits < 16) fill();
    const uint16_t e = h.look[bits >> 55];
    if (e) { const int l = e >> 8; bits <<= l; nbits -= l; return e & 0xff; }
    int code = (int)(bits >> 54), l = 10;
    for (; l <= 16; l++) {
      if (code <= h.maxcode[l]) break;
      code = (int)(bits >> (64 - l - 1));
    }
    if (l > 16) bad("JPEG: corrupt Huffman code");
    bits <<= l; nbits -= l;
    return h.vals[(code + h.valoff[l]) & 0xff];
  }
  static inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

  void build(Huff& h, const uint8_t* counts, const uint8_t* vals, int nvals) {
    memcpy(h.vals, vals, (size_t)nvals);
    memset(h.look, 0, sizeof(h.look));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      h.valoff[l] = k - code;
      for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
        if (code >= (1 << l)) bad("JPEG: bad Huffman table");
        if (l <= 9) {
          const int base = code << (9 - l);
          for (int j = 0; j < (1 << (9 - l)); j++) h.look[base + j] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      h.maxcode[l] = counts[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.ok = true;
  }

  void idct_store(const int* in, uint8_t* out, int stride) {
    // jidctint "islow": CONST_BITS 13, PASS1_BITS 2
    constexpr int CB = 13, P1 = 2;
    constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299,
                  F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int ws[64];
    auto descale = [](long x, int s) { return (int)((x + (1L << (s - 1))) >> s); };
    for (int col = 0; col < 8; col++) {
      const int* p = in + col;
      long z2 = p[16], z3 = p[48];
      long z1 = (z2 + z3) * F0541;
      long t2 = z1 + z3 * (-F1847), t3 = z1 + z2 * F0765;
      z2 = p[0]; z3 = p[32];
      long t0 = (z2 + z3) * (1L << CB), t1 = (z2 - z3) * (1L << CB);
      const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
      t0 = p[56]; t1 = p[40]; t2 = p[24]; t3 = p[8];
      z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; long z4 = t1 + t3;
      const long z5 = (z3 + z4) * F1175;
      t0 *= F0298; t1 *= F2053; t2 *= F3072; t3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
      int* w = ws + col;
      w[0] = descale(t10 + t3, CB - P1); w[56] = descale(t10 - t3, CB - P1);
      w[8] = descale(t11 + t2, CB - P1); w[48] = descale(t11 - t2, CB - P1);
      w[16] = descale(t12 + t1, CB - P1); w[40] = descale(t12 - t1, CB - P1);
      w[24] = descale(t13 + t0, CB - P1); w[32] = descale(t13 - t0, CB - P1);
    }
    for (int row = 0; row < 8; row++) {
      const int* p = ws + row * 8;
      long z2 = p[2], z3 = p[6];
      long z1 = (z2 + z3) * F0541;
      long t2 = z1 + z3 * (-F1847), t3 = z1 + z2 * F0765;
      long t0 = ((long)p[0] + p[4]) * (1L << CB), t1 = ((long)p[0] - p[4]) * (1L << CB);
      const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
      t0 = p[7]; t1 = p[5]; t2 = p[3]; t3 = p[1];
      z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; long z4 = t1 + t3;
      const long z5 = (z3 + z4) * F1175;
      t0 *= F0298; t1 *= F2053; t2 *= F3072; t3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
      uint8_t* o = out + row * stride;
      auto lim = [&](long x) { const int v = descale(x, CB + P1 + 3) + 128; return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
      o[0] = lim(t10 + t3); o[7] = lim(t10 - t3); o[1] = lim(t11 + t2); o[6] = lim(t11 - t2);
      o[2] = lim(t12 + t1); o[5] = lim(t12 - t1); o[3] = lim(t13 + t0); o[4] = lim(t13 - t0);
    }
  }

  void decode_block(JComp& cm, int bx, int by) {
    int coef[64];
    memset(coef, 0, sizeof(coef));
    const Huff& hd = dc[cm.td];
    const Huff& ha = ac[cm.ta];
    const uint16_t* q = qt[cm.tq];
    const int s = decode(hd);
    if (s > 11) bad("JPEG: corrupt DC coefficient");
    // (corrupt streams may drive the predictor or a product anywhere: wrap / clamp instead of overflowing)
    if (s) cm.pred = (int)((unsigned)cm.pred + (unsigned)extend(getbits(s), s));
    auto clampc = [](long v) { return (int)(v < -(1L << 24) ? -(1L << 24) : v > (1L << 24) ? (1L << 24) : v); };
    coef[0] = clampc((long)cm.pred * q[0]);
    bool any_ac = false;
    for (int k = 1; k < 64;) {
      const int rs = decode(ha), r = rs >> 4, sz = rs & 15;
      if (sz == 0) { if (r != 15) break; k += 16; continue; }
      k += r;
      if (k > 63) bad("JPEG: corrupt AC coefficients");
      coef[kZigzag[k]] = clampc((long)extend(getbits(sz), sz) * q[k]);
      any_ac = true;
      k++;
    }
    uint8_t* dst = cm.plane.data() + (size_t)by * 8 * cm.stride + (size_t)bx * 8;
    if (!any_ac) {  // flat block: both passes of the transform reduce to one rounding, (4 * dc + 16) >> 5
      const int v = (int)(((long)coef[0] * 4 + 16) >> 5) + 128;
      const uint8_t px = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      for (int r8 = 0; r8 < 8; r8++) memset(dst + (size_t)r8 * cm.stride, px, 8);
      return;
    }
    idct_store(coef, dst, cm.stride);
  }

  // Progressive mode (T.81 annex G): a scan carries one band [Ss, Se] of coefficients at bit position Al, either their
  // first pass (Ah = 0) or one more bit of precision (Ah > 0); the block's coefficients accumulate in cm.coef.
  void prog_block(JComp& cm, int bx, int by) {
    int* blk = cm.coef.data() + ((size_t)by * (cm.stride / 8) + bx) * 64;
    if (Ss == 0) {
      if (Ah == 0) {
        const int s = decode(dc[cm.td]);
        if (s > 11) bad("JPEG: corrupt DC coefficient");
        if (s) cm.pred = (int)((unsigned)cm.pred + (unsigned)extend(getbits(s), s));
        const long v = (long)cm.pred * (1L << Al);
        blk[0] = (int)(v < -(1L << 24) ? -(1L << 24) : v > (1L << 24) ? (1L << 24) : v);
      } else if (getbits(1)) blk[0] |= 1 << Al;
      return;
    }
    const Huff& ha = ac[cm.ta];
    if (Ah == 0) {
      if (eobrun > 0) { eobrun--; return; }
      for (int k = Ss; k <= Se; k++) {
        const int rs = decode(ha), r = rs >> 4, sz = rs & 15;
        if (sz) {
          k += r;
          if (k > 63) bad("JPEG: corrupt AC coefficients");
          blk[kZigzag[k]] = extend(getbits(sz), sz) * (1 << Al);
        } else if (r == 15) k += 15;
        else { eobrun = (1 << r) - 1; if (r) eobrun += getbits(r); break; }
      }
      return;
    }
    const int p1 = 1 << Al, m1 = -(1 << Al);
    auto correct = [&](int* co) { if (getbits(1) && (*co & p1) == 0) *co += (*co >= 0 ? p1 : m1); };
    int k = Ss;
    if (eobrun == 0) {
      for (; k <= Se; k++) {
        const int rs = decode(ha), sz = rs & 15;
        int r = rs >> 4, val = 0;
        if (sz) val = getbits(1) ? p1 : m1;  // a newly non-zero coefficient is +-1 at this bit position
        else if (r != 15) { eobrun = 1 << r; if (r) eobrun += getbits(r); break; }
        // skip r still-zero coefficients; every already non-zero one passed on the way takes a correction bit
        while (k <= Se) {
          int* co = blk + kZigzag[k];
          if (*co != 0) correct(co);
          else if (--r < 0) break;
          k++;
        }
        if (val) { if (k > 63) bad("JPEG: corrupt AC coefficients"); blk[kZigzag[k]] = val; }
      }
    }
    if (eobrun > 0) {
      for (; k <= Se; k++) { int* co = blk + kZigzag[k]; if (*co != 0) correct(co); }
      eobrun--;
    }
  }

  // progressive: after the last scan, dequantise and transform every block
  void finish_progressive() {
    for (int i = 0; i < nc; i++) {
      JComp& cm = c[i];
      if (!qt_ok[cm.tq]) bad("JPEG: frame refers to a missing quantisation table");
      int qn[64];
      for (int k = 0; k < 64; k++) qn[kZigzag[k]] = qt[cm.tq][k];
      const int bw = cm.stride / 8, bh = cm.rows / 8;
      for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
          const int* blk = cm.coef.data() + ((size_t)by * bw + bx) * 64;
          int coef[64];
          bool any_ac = false;
          for (int k = 0; k < 64; k++) {
            const long v = (long)blk[k] * qn[k];
            coef[k] = (int)(v < -(1L << 24) ? -(1L << 24) : v > (1L << 24) ? (1L << 24) : v);
            any_ac |= k > 0 && coef[k] != 0;
          }
          uint8_t* dst = cm.plane.data() + (size_t)by * 8 * cm.stride + (size_t)bx * 8;
          if (!any_ac) {
            const int v = (int)(((long)coef[0] * 4 + 16) >> 5) + 128;
            const uint8_t px = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            for (int r8 = 0; r8 < 8; r8++) memset(dst + (size_t)r8 * cm.stride, px, 8);
          } else idct_store(coef, dst, cm.stride);
        }
    }
  }

  void restart_marker(int& expect) {
    // discard the remaining bits, then the next two bytes must be RSTn
    bits = 0; nbits = 0;
    if (hit_marker) hit_marker = false;
    while (pos < n && d[pos] != 0xff) pos++;   // tolerate fill bytes before the marker
    while (pos + 1 < n && d[pos + 1] == 0xff) pos++;
    if (pos + 1 >= n || d[pos + 1] != (0xd0 | expect)) bad("JPEG: missing restart marker");
    pos += 2;
    expect = (expect + 1) & 7;
    for (int i = 0; i < nc; i++) c[i].pred = 0;
    eobrun = 0;
  }

  void scan(const int* idx, int ns) {
    for (int i = 0; i < ns; i++) {
      JComp& cm = c[idx[i]];
      const bool need_dc = !progressive || (Ss == 0 && Ah == 0), need_ac = !progressive || Ss > 0;
      if ((need_dc && !dc[cm.td].ok) || (need_ac && !ac[cm.ta].ok)) bad("JPEG: scan refers to a missing Huffman table");
      if (!progressive && !qt_ok[cm.tq]) bad("JPEG: frame refers to a missing quantisation table");
      cm.pred = 0;
    }
    if (progressive) {
      if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) bad("JPEG: bad progressive scan parameters");
    }
    eobrun = 0;
    bits = 0; nbits = 0; hit_marker = false;
    int expect = 0;
    long count = 0;
    if (ns == 1) {  // non-interleaved: the component's own blocks in raster order
      JComp& cm = c[idx[0]];
      const int bw = (cm.cw + 7) / 8, bh = (cm.ch + 7) / 8;
      for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
          if (restart && count && count % restart == 0) restart_marker(expect);
          if (progressive) prog_block(cm, bx, by); else decode_block(cm, bx, by);
          count++;
        }
    } else {
      const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
      for (int y = 0; y < my; y++)
        for (int x = 0; x < mx; x++) {
          if (restart && count && count % restart == 0) restart_marker(expect);
          for (int i = 0; i < ns; i++) {
            JComp& cm = c[idx[i]];
            for (int v = 0; v < cm.vs; v++)
              for (int h = 0; h < cm.hs; h++) {
                if (progressive) prog_block(cm, x * cm.hs + h, y * cm.vs + v); else decode_block(cm, x * cm.hs + h, y * cm.vs + v);
              }
          }
          count++;
        }
    }
    // resume marker parsing after the entropy-coded data: pos sits on the 0xff of the marker that stopped the reader
    if (!hit_marker) { while (pos + 1 < n && !(d[pos] == 0xff && d[pos + 1] != 0x00 && d[pos + 1] != 0xff)) pos++; }
  }

  // One component plane -> full resolution [H][W]
  void upsample(const JComp& cm, std::vector<uint8_t>& out) {
    out.resize((size_t)W * H);
    const int fh = hmax / cm.hs, fv = vmax / cm.vs;
    const bool exact = hmax % cm.hs == 0 && vmax % cm.vs == 0;
    if (!exact) bad("JPEG: fractional sampling ratios are not supported");
    const int cw = cm.cw, ch = cm.ch;
    auto row = [&](int y) { return cm.plane.data() + (size_t)std::min(std::max(y, 0), ch - 1) * cm.stride; };
    if (fh == 1 && fv == 1) {
      for (int y = 0; y < H; y++) memcpy(out.data() + (size_t)y * W, row(y), (size_t)W);
    } else if (fh == 2 && fv == 1 && cw > 2) {
      std::vector<uint8_t> line((size_t)cw * 2);
      for (int y = 0; y < H; y++) {
        const uint8_t* in = row(y);
        line[0] = in[0]; line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        for (int i = 1; i < cw - 1; i++) {
          const int v = in[i] * 3;
          line[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2); line[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
        }
        line[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2); line[2 * cw - 1] = in[cw - 1];
        memcpy(out.data() + (size_t)y * W, line.data(), (size_t)W);
      }
    } else if (fh == 2 && fv == 2 && cw > 2) {
      std::vector<uint8_t> line((size_t)cw * 2);
      std::vector<int> sum((size_t)cw);
      for (int y = 0; y < H; y++) {
        const int r = y >> 1;
        const uint8_t* in0 = row(r);
        const uint8_t* in1 = row((y & 1) ? r + 1 : r - 1);
        for (int i = 0; i < cw; i++) sum[i] = in0[i] * 3 + in1[i];
        line[0] = (uint8_t)((sum[0] * 4 + 8) >> 4); line[1] = (uint8_t)((sum[0] * 3 + sum[1] + 7) >> 4);
        for (int i = 1; i < cw - 1; i++) {
          line[2 * i] = (uint8_t)((sum[i] * 3 + sum[i - 1] + 8) >> 4); line[2 * i + 1] = (uint8_t)((sum[i] * 3 + sum[i + 1] + 7) >> 4);
        }
        line[2 * cw - 2] = (uint8_t)((sum[cw - 1] * 3 + sum[cw - 2] + 8) >> 4); line[2 * cw - 1] = (uint8_t)((sum[cw - 1] * 4 + 7) >> 4);
        memcpy(out.data() + (size_t)y * W, line.data(), (size_t)W);
      }
    } else if (fh == 1 && fv == 2) {
      for (int y = 0; y < H; y++) {
        const int r = y >> 1, bias = (y & 1) ? 2 : 1;
        const uint8_t* in0 = row(r);
        const uint8_t* in1 = row((y & 1) ? r + 1 : r - 1);
        uint8_t* o = out.data() + (size_t)y * W;
        for (int x = 0; x < W; x++) o[x] = (uint8_t)((in0[x] * 3 + in1[x] + bias) >> 2);
      }
    } else {  // any other integer factors: replication
      for (int y = 0; y < H; y++) {
        const uint8_t* in = row(y / fv);
        uint8_t* o = out.data() + (size_t)y * W;
        for (int x = 0; x < W; x++) o[x] = in[std::min(x / fh, cw - 1)];
      }
    }
  }

  void run(std::vector<uint8_t>* rgb, int* oh, int* ow) {
    bool done = false;
    int scans = 0;
    while (!done) {
      // next marker
      while (pos < n && d[pos] != 0xff) pos++;
      while (pos < n && d[pos] == 0xff) pos++;
      if (pos >= n) { if (scans) break; bad("JPEG: no image data before the end of the file"); }
      const int m = d[pos++];
      if (m == 0xd9) break;                                   // EOI
      if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;    // TEM / stray RSTn: no payload
      if (pos + 2 > n) bad("JPEG: truncated marker segment");
      const size_t L = be16(d + pos);
      if (L < 2 || pos + L > n) bad("JPEG: marker segment past the end of the data");
      const uint8_t* s = d + pos + 2;
      const size_t sl = L - 2;
      pos += L;
      if (m == 0xdb) {  // DQT
        size_t i = 0;
        while (i < sl) {
          const int pq = s[i] >> 4, tq = s[i] & 15;
          i++;
          if (tq > 3 || pq > 1 || i + (size_t)64 * (pq + 1) > sl) bad("JPEG: bad quantisation table");
          for (int k = 0; k < 64; k++) { qt[tq][k] = (uint16_t)(pq ? be16(s + i + 2 * k) : s[i + k]); }
          i += (size_t)64 * (pq + 1);
          qt_ok[tq] = true;
        }
      } else if (m == 0xc4) {  // DHT
        size_t i = 0;
        while (i < sl) {
          if (i + 17 > sl) bad("JPEG: bad Huffman table");
          const int tc = s[i] >> 4, th = s[i] & 15;
          int total = 0;
          for (int k = 0; k < 16; k++) total += s[i + 1 + k];
          if (tc > 1 || th > 3 || total > 256 || i + 17 + (size_t)total > sl) bad("JPEG: bad Huffman table");
          build(tc ? ac[th] : dc[th], s + i + 1, s + i + 17, total);
          i += 17 + (size_t)total;
        }
      } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {  // SOF0 / SOF1 (sequential), SOF2 (progressive)
        if (have_sof) bad("JPEG: more than one frame header");
        if (sl < 6) bad("JPEG: bad frame header");
        if (s[0] != 8) bad("JPEG: only 8-bit samples are supported");
        H = (int)be16(s + 1); W = (int)be16(s + 3); nc = s[5];
        check_dims((uint64_t)W, (uint64_t)H);
        if (nc == 4) bad("JPEG: CMYK / four-component files are not supported");
        if ((nc != 1 && nc != 3) || sl < 6 + (size_t)3 * nc) bad("JPEG: bad component count");
        for (int i = 0; i < nc; i++) {
          c[i].id = s[6 + 3 * i]; c[i].hs = s[7 + 3 * i] >> 4; c[i].vs = s[7 + 3 * i] & 15; c[i].tq = s[8 + 3 * i];
          if (c[i].hs < 1 || c[i].hs > 4 || c[i].vs < 1 || c[i].vs > 4 || c[i].tq > 3) bad("JPEG: bad sampling factors");
          hmax = std::max(hmax, c[i].hs); vmax = std::max(vmax, c[i].vs);
        }
        if (nc == 1) { c[0].hs = c[0].vs = 1; hmax = vmax = 1; }  // a single component is never subsampled
        const int mx = (W + 8 * hmax - 1) / (8 * hmax), my = (H + 8 * vmax - 1) / (8 * vmax);
        for (int i = 0; i < nc; i++) {
          c[i].cw = (W * c[i].hs + hmax - 1) / hmax; c[i].ch = (H * c[i].vs + vmax - 1) / vmax;
          c[i].stride = mx * c[i].hs * 8; c[i].rows = my * c[i].vs * 8;
          c[i].plane.assign((size_t)c[i].stride * c[i].rows, 128);
          if (m == 0xc2) c[i].coef.assign((size_t)c[i].stride * c[i].rows, 0);
        }
        progressive = m == 0xc2;
        have_sof = true;
      }
      else if (m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) bad("JPEG: lossless / hierarchical / arithmetic-coded files are not supported");
      else if (m == 0xdd) { if (sl < 2) bad("JPEG: bad DRI"); restart = (int)be16(s); }
      else if (m == 0xee) { if (sl >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_tf = s[11]; } }
      else if (m == 0xda) {  // SOS
        if (!have_sof) bad("JPEG: scan before the frame header");
        if (sl < 1) bad("JPEG: bad scan header");
        const int ns = s[0];
        if (ns < 1 || ns > nc || sl < 1 + (size_t)2 * ns + 3) bad("JPEG: bad scan header");
        int idx[3];
        for (int i = 0; i < ns; i++) {
          int k = -1;
          for (int j = 0; j < nc; j++) if (c[j].id == s[1 + 2 * i]) k = j;
          if (k < 0) bad("JPEG: scan names an unknown component");
          idx[i] = k;
          c[k].td = s[2 + 2 * i] >> 4; c[k].ta = s[2 + 2 * i] & 15;
          if (c[k].td > 3 || c[k].ta > 3) bad("JPEG: bad table selector");
        }
        if (ns > 1 && ns != nc) bad("JPEG: partially interleaved scans are not supported");
        const uint8_t* sp = s + 1 + 2 * ns;
        Ss = sp[0]; Se = sp[1]; Ah = sp[2] >> 4; Al = sp[2] & 15;
        scan(idx, ns);
        scans++;
      }
      // everything else (APPn, COM, ...) is skipped
    }
    if (!have_sof || !scans) bad("JPEG: no image data");
    if (progressive) finish_progressive();
    rgb->assign((size_t)W * H * 3, 0);
    if (nc == 1) {
      for (int y = 0; y < H; y++) {
        const uint8_t* in = c[0].plane.data() + (size_t)y * c[0].stride;
        uint8_t* o = rgb->data() + (size_t)y * W * 3;
        for (int x = 0; x < W; x++) o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = in[x];
      }
    } else {
      std::vector<uint8_t> p[3];
      for (int i = 0; i < 3; i++) upsample(c[i], p[i]);
      const bool is_rgb = adobe ? adobe_tf == 0 : (c[0].id == 'R' && c[1].id == 'G' && c[2].id == 'B');
      const size_t np = (size_t)W * H;
      uint8_t* o = rgb->data();
      if (is_rgb) {
        for (size_t i = 0; i < np; i++) { o[3 * i] = p[0][i]; o[3 * i + 1] = p[1][i]; o[3 * i + 2] = p[2][i]; }
      } else {
        // jdcolor: 16-bit fixed point, FIX(x) = (int)(x * 65536 + 0.5)
        int crr[256], cbb[256]; long crg[256], cbg[256];
        for (int i = 0; i < 256; i++) {
          const long x = i - 128;
          crr[i] = (int)((91881L * x + 32768) >> 16); cbb[i] = (int)((116130L * x + 32768) >> 16);
          crg[i] = -46802L * x; cbg[i] = -22554L * x + 32768;
        }
        auto lim = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
        for (size_t i = 0; i < np; i++) {
          const int yv = p[0][i], cb = p[1][i], cr = p[2][i];
          o[3 * i] = lim(yv + crr[cr]); o[3 * i + 1] = lim(yv + (int)((cbg[cb] + crg[cr]) >> 16)); o[3 * i + 2] = lim(yv + cbb[cb]);
        }
      }
    }
    *oh = H; *ow = W;
  }
};

}  // namespace

void decode_image(const uint8_t* data, size_t len, std::vector<uint8_t>* rgb, int* h, int* w) {
  static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (!data || len < 4) bad("empty or truncated input");
  if (len >= 8 && !memcmp(data, png_sig, 8)) return decode_png(data, len, rgb, h, w);
  if (data[0] == 0xff && data[1] == 0xd8) { JpegDec j{data, len}; return j.run(rgb, h, w); }
  if (data[0] == 'P' && (data[1] == '2' || data[1] == '3' || data[1] == '5' || data[1] == '6')) return decode_pnm(data, len, rgb, h, w);
  if (data[0] == 'B' && data[1] == 'M') return decode_bmp(data, len, rgb, h, w);
  // Formats image 0.25.6 reads with its default features (/root/reference/Cargo.toml:23) that this decoder does not: named,
  // so that a caller sees WHICH decoder is missing instead of the reference's "format could not be determined".
  struct Sig { const char* magic; size_t n; size_t off; const char* name; };
  static const Sig sigs[] = {{"GIF87a", 6, 0, "GIF"}, {"GIF89a", 6, 0, "GIF"}, {"WEBP", 4, 8, "WebP"}, {"II*\0", 4, 0, "TIFF"}, {"MM\0*", 4, 0, "TIFF"},
                             {"\x76\x2f\x31\x01", 4, 0, "OpenEXR"}, {"#?RADIANCE", 10, 0, "Radiance HDR"}, {"qoif", 4, 0, "QOI"},
                             {"\0\0\1\0", 4, 0, "ICO"}, {"DDS ", 4, 0, "DDS"}, {"farbfeld", 8, 0, "farbfeld"}, {"ftypavif", 8, 4, "AVIF"}};
  for (const Sig& g : sigs)
    if (len >= g.off + g.n && !memcmp(data + g.off, g.magic, g.n))
      bad(std::string(g.name) + " input: the reference decodes it (image crate default features), this decoder reads PNG, JPEG, PNM and BMP only");
  bad("unrecognised image format (PNG, JPEG, PNM and BMP are read)");
}

}  // namespace rt
