/*
 * orc_synth.c -- synthetic 640x480 luma frames / 428x270 card crops for tests
 * and the benchmark (test infrastructure; the HIP twin that fills HBM-resident
 * batches is card.io-dmz_amd/csrc/synth.hip and must produce identical bytes).
 *
 * Not derived from the reference (it ships no test images); the scene follows
 * noisy dark background, a bright card quad whose corners are
 * the guide-frame corners (dmz_constants.h:16-27 => (106,105)-(533,374)) plus a
 * per-corner jitter, 16 Luhn-valid digits in the 4-4-4-4 layout -- or, for every
 * tenth card or so, 15 digits in the 4-6-5 layout with prefix 34 / 37 (the second
 * number pattern of n_vseg.cpp:26-31 / n_hseg.cpp) -- rendered as soft-edged
 * 7-segment strokes in the number band of the card.
 *
 * Determinism contract: per-frame parameters come from splitmix64(seed, frame);
 * per-pixel noise from a 32-bit integer hash; the frame->card mapping is IEEE
 * double (+,*,/ only, no contraction); everything after that is integer.
 */
#include "dmz_oracle.h"

#include <math.h>
#include <string.h>

typedef struct {
  double h[9];       /* frame (x,y,1) -> card (u,v,w) homography */
  int bg, card;      /* mean grey levels */
  int ink;           /* digit stroke grey delta (signed) */
  int rim;           /* emboss rim delta */
  int noise_bg, noise_card; /* noise amplitude, 1/256 units of the ~+-510 sum */
  int x0, y0;        /* card-space origin of the first digit box, 1/16 px */
  int pitch;         /* digit pitch in 1/16 px */
  uint8_t digits[16];
  int ex0, ey0;      /* card-space origin of the MM/YY line, 1/16 px */
  int epitch;        /* its character pitch, 1/16 px */
  uint8_t exp_digits[4]; /* M M Y Y */
  int kind;          /* 0: 16 digits 4-4-4-4; 1: 15 digits 4-6-5 */
} synth_params;

static uint64_t splitmix64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint32_t hash32(uint32_t a) {
  a ^= a >> 16; a *= 0x7feb352du;
  a ^= a >> 15; a *= 0x846ca68bu;
  a ^= a >> 16;
  return a;
}

static int noise_at(uint32_t key, int x, int y) { /* ~triangular/gaussian, range +-510 */
  uint32_t h = hash32(key ^ hash32((uint32_t)(y * 1024 + x) + 0x9e3779b9u));
  return (int)(h & 255) + (int)((h >> 8) & 255) + (int)((h >> 16) & 255) + (int)(h >> 24) - 510;
}

/* unit square (0,0),(1,0),(0,1),(1,1) scaled to the card rect -> quad; returns the
 * inverse (frame -> card) as a plain adjugate (scale-free, w divides it out). */
static void quad_homography(const double q[8] /* tl,tr,bl,br */, double hinv[9]) {
  const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
  /* Heckbert square->quad with corners ordered (0,0)=tl,(1,0)=tr,(1,1)=br,(0,1)=bl */
  const double dx1 = x1 - x3, dx2 = x2 - x3, dx3 = x0 - x1 + x3 - x2;
  const double dy1 = y1 - y3, dy2 = y2 - y3, dy3 = y0 - y1 + y3 - y2;
  const double den = dx1 * dy2 - dx2 * dy1;
  const double g = (dx3 * dy2 - dx2 * dy3) / den;
  const double hh = (dx1 * dy3 - dx3 * dy1) / den;
  /* card(u in [0,427], v in [0,269]) -> frame */
  const double su = 1.0 / 427.0, sv = 1.0 / 269.0;
  double m[9];
  m[0] = (x1 - x0 + g * x1) * su; m[1] = (x2 - x0 + hh * x2) * sv; m[2] = x0;
  m[3] = (y1 - y0 + g * y1) * su; m[4] = (y2 - y0 + hh * y2) * sv; m[5] = y0;
  m[6] = g * su;                  m[7] = hh * sv;                  m[8] = 1.0;
  hinv[0] = m[4] * m[8] - m[5] * m[7];
  hinv[1] = m[2] * m[7] - m[1] * m[8];
  hinv[2] = m[1] * m[5] - m[2] * m[4];
  hinv[3] = m[5] * m[6] - m[3] * m[8];
  hinv[4] = m[0] * m[8] - m[2] * m[6];
  hinv[5] = m[2] * m[3] - m[0] * m[5];
  hinv[6] = m[3] * m[7] - m[4] * m[6];
  hinv[7] = m[1] * m[6] - m[0] * m[7];
  hinv[8] = m[0] * m[4] - m[1] * m[3];
}

static void make_params(uint64_t seed, uint64_t frame, synth_params *p) {
  uint64_t s0 = seed ^ 0xCA4D10ull, s1 = frame;
  uint64_t s = splitmix64(&s0) ^ (splitmix64(&s1) * 0xD1342543DE82EF95ull); /* decorrelate frames */
  double q[8];
  static const int base[8] = {106, 105, 533, 105, 106, 374, 533, 374}; /* tl,tr,bl,br */
  for (int i = 0; i < 8; i++) {
    int j = (int)(splitmix64(&s) % 193) - 96; /* +-6 px in 1/16 px */
    q[i] = (double)base[i] + (double)j * 0.0625;
  }
  quad_homography(q, p->h);
  p->bg = 56 + (int)(splitmix64(&s) % 17);
  p->card = 168 + (int)(splitmix64(&s) % 17);
  p->ink = -96;
  p->rim = 40;
  p->noise_bg = 6;
  p->noise_card = 3;
  p->x0 = 40 * 16 + (int)(splitmix64(&s) % 65) - 32;
  p->y0 = 151 * 16 + (int)(splitmix64(&s) % 129) - 64;
  p->pitch = 293; /* 18.3125 px */
  /* 16 digits, first one 4 (Visa), last chosen so that the Luhn sum is 0 mod 10 */
  int sum = 0;
  p->digits[0] = 4;
  for (int i = 1; i < 15; i++) p->digits[i] = (uint8_t)(splitmix64(&s) % 10);
  for (int i = 0; i < 15; i++) {
    int d = p->digits[i];
    if ((i & 1) == 0) { d *= 2; d = d % 10 + d / 10; } /* positions doubled for a 16-digit PAN */
    sum += d;
  }
  p->digits[15] = (uint8_t)((10 - sum % 10) % 10);
  /* expiry line "MM/YY": small characters (9 x 15 px boxes) below the number */
  p->ex0 = 190 * 16 + (int)(splitmix64(&s) % 257) - 128;
  p->ey0 = 206 * 16 + (int)(splitmix64(&s) % 129) - 64;
  p->epitch = 13 * 16 + (int)(splitmix64(&s) % 17) - 8;
  const int month = 1 + (int)(splitmix64(&s) % 12), year = 27 + (int)(splitmix64(&s) % 4);
  p->exp_digits[0] = (uint8_t)(month / 10);
  p->exp_digits[1] = (uint8_t)(month % 10);
  p->exp_digits[2] = (uint8_t)(year / 10);
  p->exp_digits[3] = (uint8_t)(year % 10);
  /* the card kind is the LAST draw, so that the 16-digit cards keep the bytes they had before the kind existed */
  p->kind = (splitmix64(&s) % 10) == 0;
  if (p->kind) {
    /* 15 digits, prefix 34 / 37; Luhn: from the right every second digit doubled = the odd indices of a 15-digit PAN */
    p->digits[0] = 3;
    p->digits[1] = (p->digits[1] & 1) ? 7 : 4;
    sum = 0;
    for (int i = 0; i < 14; i++) {
      int d = p->digits[i];
      if (i & 1) { d *= 2; d = d % 10 + d / 10; }
      sum += d;
    }
    p->digits[14] = (uint8_t)((10 - sum % 10) % 10);
    p->digits[15] = 0;
  }
}

/* 7 segments in a 17 x 25 px digit box, 1/16 px units: x0,y0,x1,y1 */
static const int16_t k_seg[7][4] = {
    {2 * 16, 0 * 16, 15 * 16, 3 * 16},    /* a top */
    {14 * 16, 1 * 16, 17 * 16, 13 * 16},  /* b top right */
    {14 * 16, 12 * 16, 17 * 16, 24 * 16}, /* c bottom right */
    {2 * 16, 22 * 16, 15 * 16, 25 * 16},  /* d bottom */
    {0 * 16, 12 * 16, 3 * 16, 24 * 16},   /* e bottom left */
    {0 * 16, 1 * 16, 3 * 16, 13 * 16},    /* f top left */
    {2 * 16, 11 * 16, 15 * 16, 14 * 16},  /* g middle */
};
/* the same seven segments in a 9 x 15 px box (expiry characters), stroke 2 px */
static const int16_t k_seg_small[7][4] = {
    {1 * 16, 0 * 16, 8 * 16, 2 * 16},    /* a */
    {7 * 16, 1 * 16, 9 * 16, 8 * 16},    /* b */
    {7 * 16, 7 * 16, 9 * 16, 14 * 16},   /* c */
    {1 * 16, 13 * 16, 8 * 16, 15 * 16},  /* d */
    {0 * 16, 7 * 16, 2 * 16, 14 * 16},   /* e */
    {0 * 16, 1 * 16, 2 * 16, 8 * 16},    /* f */
    {1 * 16, 104, 8 * 16, 136},          /* g (6.5 .. 8.5 px) */
};
static const uint8_t k_digit_segs[10] = {
    0x3F, 0x06, 0x5B, 0x4F, 0x66, 0x6D, 0x7D, 0x07, 0x7F, 0x6F};

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* soft coverage (0..256) of point (px,py) [1/16 px, box-local] by the digit's strokes,
 * evaluated for a box offset (ox,oy) so that the emboss rim can reuse it */
static int stroke_cov_tab(const int16_t seg[7][4], int segs, int px, int py) {
  int best = 0;
  for (int s = 0; s < 7; s++) {
    if (!(segs & (1 << s))) continue;
    int cx = clampi((px - seg[s][0] < seg[s][2] - px ? px - seg[s][0] : seg[s][2] - px) + 8, 0, 16);
    int cy = clampi((py - seg[s][1] < seg[s][3] - py ? py - seg[s][1] : seg[s][3] - py) + 8, 0, 16);
    int c = cx * cy;
    if (c > best) best = c;
  }
  return best; /* 0..256 */
}
static int stroke_cov(int segs, int px, int py) { return stroke_cov_tab(k_seg, segs, px, py); }

/* '/' in the 9 x 15 box: a 2 px wide stroke from (1.5, 15) to (7.5, 0) */
static int slash_cov(int px, int py) {
  const int xl = 24 + ((240 - py) * 96) / 240;
  const int dx = px - xl < 0 ? xl - px : px - xl;
  const int cx = clampi(16 - dx + 8, 0, 16);
  const int cy = clampi((py < 240 - py ? py : 240 - py) + 8, 0, 16);
  return cx * cy;
}

/* card texture at card-space point (U,V) in 1/16 px; returns grey delta vs card mean */
static int card_delta(const synth_params *p, int U, int V) {
  int d = 0;
  /* low-frequency shading: +-4 triangle wave across the card */
  int t = (U >> 4) & 127;
  d += ((t < 64 ? t : 127 - t) - 32) >> 3;
  int ry = V - p->y0;
  if (ry >= -32 && ry < 25 * 16 + 32) {
    int rx = U - p->x0;
    if (rx >= -32) {
      /* 4-4-4-4 layout: slot index over 19 slots, slots 4, 9, 14 are gaps; 4-6-5: 17 slots, gaps at 4 and 11 */
      int slot = (rx + 32) / p->pitch;
      if (p->kind ? (slot < 17 && slot != 4 && slot != 11) : (slot < 19 && (slot % 5) != 4)) {
        int di = p->kind ? slot - (slot > 4) - (slot > 11) : slot - slot / 5;
        int lx = rx - slot * p->pitch;
        int segs = k_digit_segs[p->digits[di]];
        int c0 = stroke_cov(segs, lx, ry);
        int c1 = stroke_cov(segs, lx + 16, ry + 16); /* highlight up-left of the stroke */
        int c2 = stroke_cov(segs, lx - 16, ry - 16); /* shadow down-right of the stroke */
        d += (p->ink * c0) >> 8;
        d += (p->rim * (c1 - c0 > 0 ? c1 - c0 : 0)) >> 8;
        d -= (p->rim * (c2 - c0 > 0 ? c2 - c0 : 0)) >> 8;
      }
    }
  }
  int ey = V - p->ey0;
  if (ey >= -32 && ey < 15 * 16 + 32) {
    int ex = U - p->ex0;
    if (ex >= -32) {
      int slot = (ex + 32) / p->epitch;
      if (slot < 5) {
        int lx = ex - slot * p->epitch;
        int c0, c1, c2;
        if (slot == 2) {
          c0 = slash_cov(lx, ey);
          c1 = slash_cov(lx + 16, ey + 16);
          c2 = slash_cov(lx - 16, ey - 16);
        } else {
          int segs = k_digit_segs[p->exp_digits[slot > 2 ? slot - 1 : slot]];
          c0 = stroke_cov_tab(k_seg_small, segs, lx, ey);
          c1 = stroke_cov_tab(k_seg_small, segs, lx + 16, ey + 16);
          c2 = stroke_cov_tab(k_seg_small, segs, lx - 16, ey - 16);
        }
        d += (p->ink * c0) >> 8;
        d += (p->rim * (c1 - c0 > 0 ? c1 - c0 : 0)) >> 8;
        d -= (p->rim * (c2 - c0 > 0 ? c2 - c0 : 0)) >> 8;
      }
    }
  }
  return d;
}

void orc_synth_frame(uint64_t seed, uint64_t frame, uint8_t *y, uint8_t digits_out[16]) {
  synth_params p;
  make_params(seed, frame, &p);
  const uint32_t key = hash32((uint32_t)(seed * 0x9E3779B1u) ^ hash32((uint32_t)frame) ^ (uint32_t)(frame >> 32));
  for (int yy = 0; yy < 480; yy++) {
    for (int xx = 0; xx < 640; xx++) {
      const double fx = (double)xx, fy = (double)yy;
      const double w = (p.h[6] * fx + p.h[7] * fy) + p.h[8];
      const double u = ((p.h[0] * fx + p.h[1] * fy) + p.h[2]) / w;
      const double v = ((p.h[3] * fx + p.h[4] * fy) + p.h[5]) / w;
      const int n = noise_at(key, xx, yy);
      int val;
      /* soft card boundary: coverage from the distance to the card rect in card space */
      const int U = (int)floor(u * 16.0), V = (int)floor(v * 16.0);
      int cu = clampi((U < 427 * 16 - U ? U : 427 * 16 - U) + 8, 0, 16);
      int cv = clampi((V < 269 * 16 - V ? V : 269 * 16 - V) + 8, 0, 16);
      if (u < -4.0 || u > 431.0 || v < -4.0 || v > 273.0) cu = 0;
      const int cov = cu * cv; /* 0..256 */
      const int bgv = p.bg + ((n * p.noise_bg) >> 8);
      if (cov == 0) {
        val = bgv;
      } else {
        const int cardv = p.card + card_delta(&p, U, V) + ((n * p.noise_card) >> 8);
        val = (bgv * (256 - cov) + cardv * cov + 128) >> 8;
      }
      y[yy * 640 + xx] = (uint8_t)clampi(val, 0, 255);
    }
  }
  if (digits_out) memcpy(digits_out, p.digits, 16);
}

void orc_synth_card(uint64_t seed, uint64_t frame, uint8_t *card, uint8_t digits_out[16]) {
  synth_params p;
  make_params(seed, frame, &p);
  const uint32_t key = hash32((uint32_t)(seed * 0x9E3779B1u) ^ hash32((uint32_t)frame) ^ (uint32_t)(frame >> 32)) ^ 0x5bd1e995u;
  for (int v = 0; v < 270; v++)
    for (int u = 0; u < 428; u++) {
      const int n = noise_at(key, u, v);
      int val = p.card + card_delta(&p, u * 16, v * 16) + ((n * p.noise_card) >> 8);
      card[v * 428 + u] = (uint8_t)clampi(val, 0, 255);
    }
  if (digits_out) memcpy(digits_out, p.digits, 16);
}
