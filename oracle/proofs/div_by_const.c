/* Exhaustive check (test infrastructure, not shipped): for a float32 constant b, is
 *     q  = a * y;            y = fl32(1 / b)
 *     r  = fma(-q, b, a);
 *     q' = fma(r, y, q);
 * bit-identical to the IEEE-754 float32 division a / b for EVERY float32 a?
 * (Markstein-style division by a constant; the shifting-baseline kernel uses it for "/ smooth_days" and
 * "/ window_year_baseline" in place of the ~11-instruction v_div_scale / v_div_fmas / v_div_fixup sequence.)
 * Prints, per divisor, the number of inputs where the two differ, split by class of a, plus the smallest and
 * largest |a| that differ.  NaN results compare equal to NaN results.
 *
 *   gcc -O2 -mfma -fopenmp -ffp-contract=off oracle/proofs/div_by_const.c -o oracle/proofs/_build/div_by_const -lm
 *   oracle/proofs/_build/div_by_const 1 64
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
    int b0 = argc > 1 ? atoi(argv[1]) : 1, b1 = argc > 2 ? atoi(argv[2]) : 64;
    int total_bad = 0;
    for (int bi = b0; bi <= b1; ++bi) {
        const float b = (float)bi;
        volatile float one = 1.0f;
        const float y = one / b;
        uint64_t bad_fin = 0, bad_inf = 0, bad_sub = 0;
        uint32_t min_bad = 0xFFFFFFFFu, max_bad = 0;
#pragma omp parallel for reduction(+ : bad_fin, bad_inf, bad_sub) reduction(min : min_bad) reduction(max : max_bad) schedule(static)
        for (int64_t i = 0; i < (1ll << 32); ++i) {
            const uint32_t u = (uint32_t)i;
            const float a = u2f(u);
            const float q = a * y;
            const float r = fmaf(-q, b, a);
            const float q2 = fmaf(r, y, q);
            const float ref = a / b;
            if (f2u(q2) != f2u(ref) && !(q2 != q2 && ref != ref)) {
                const uint32_t mag = u & 0x7FFFFFFFu;
                if (mag == 0x7F800000u) ++bad_inf;
                else if (fabsf(ref) < 1.17549435e-38f) ++bad_sub;
                else ++bad_fin;
                if (mag < min_bad) min_bad = mag;
                if (mag > max_bad) max_bad = mag;
            }
        }
        printf("b=%2d  y=%a  differ: normal-result=%llu  subnormal-result=%llu  a=inf=%llu  |a| range [%g, %g]\n", bi,
               (double)y, (unsigned long long)bad_fin, (unsigned long long)bad_sub, (unsigned long long)bad_inf,
               min_bad == 0xFFFFFFFFu ? 0.0 : (double)u2f(min_bad), (double)u2f(max_bad));
        fflush(stdout);
        total_bad += (bad_fin != 0);
    }
    return total_bad ? 1 : 0;
}
