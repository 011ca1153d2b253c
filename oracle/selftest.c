/* Sanitizer self-test of the CPU oracle: Random123 KATs + a small cross under ASan/UBSan. */
#include "breed_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(void) {
  uint32_t a, b;
  bo_threefry2x32(0, 0, 0, 0, &a, &b);
  CHECK(a == 0x6b200159u && b == 0x99ba4efeu);
  bo_threefry2x32(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, &a, &b);
  CHECK(a == 0x1cb996fcu && b == 0xbb002be7u);
  bo_threefry2x32(0x13198a2eu, 0x03707344u, 0x243f6a88u, 0x85a308d3u, &a, &b);
  CHECK(a == 0xc4923a9cu && b == 0x483df7a0u);

  enum { E = 2, NP = 5, NO = 3, M = 37, T = 2 };
  uint8_t pool[E * NP * M * 2], out[E * NO * M * 2], masks[E * NO * 2 * M];
  int32_t idx[E * NO * 2];
  float r[M], eff[M * T], g[E * NO * T];
  srand(1);
  for (size_t i = 0; i < sizeof pool; ++i) pool[i] = rand() & 1;
  for (int i = 0; i < E * NO * 2; ++i) idx[i] = rand() % NP;
  for (int m = 0; m < M; ++m) { r[m] = (m % 9 == 0) ? 0.5f : 0.05f; eff[2 * m] = m; eff[2 * m + 1] = -m; }
  uint32_t key[2];
  bo_key_from_seed(7, key);
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(bo_cross(pool, E, NP, idx, NO, M, r, key, mode, out, masks) == 0);
    for (int c = 0; c < E * NO; ++c)
      for (int m = 0; m < M; ++m)
        for (int p = 0; p < 2; ++p) {
          const uint8_t *par = pool + ((c / NO) * NP + idx[2 * c + p]) * 2 * M;
          CHECK(out[(c * M + m) * 2 + p] == par[2 * m + masks[(2 * c + p) * M + m]]);
        }
  }
  bo_gebv(out, E * NO, M, eff, T, g, NULL);
  int32_t f[E * 3];
  bo_founders(key, E, NP, 3, f);
  for (int i = 0; i < E * 3; ++i) CHECK(f[i] >= 0 && f[i] < NP);
  CHECK(bo_norm_index(NP, NP) == NP - 1 && bo_norm_index(-1, NP) == NP - 1 && bo_norm_index(-NP - 3, NP) == 0);
  printf(fails ? "selftest: %d failures\n" : "selftest: ok\n", fails);
  return fails != 0;
}
