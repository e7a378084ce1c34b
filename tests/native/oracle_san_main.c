/* oracle_san_main.c -- the CPU oracle under AddressSanitizer / UndefinedBehaviorSanitizer (built by
 * `make -C oracle SAN=1`; run by tests/test_cpu_sanitizers.py): the whole pipeline on a 640x480 value-noise frame and on
 * the edge sizes the parity tests use (planes narrower than the widest kernel, one-pixel dimensions in the last
 * octaves, odd sizes), both detectors, plus the matcher and the text writer.  Prints the feature counts. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sift_oracle.h"

int main(int argc, char **argv)
{
    static const int sizes[][2] = {{640, 480}, {8, 8}, {9, 33}, {33, 9}, {17, 16}, {64, 48}, {101, 67}, {2, 2}, {31, 200}};
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    size_t total = 0;
    so_keypoint *prev = NULL;
    size_t n_prev = 0;
    for (size_t s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
        const int w = sizes[s][0], h = sizes[s][1];
        uint8_t *img = (uint8_t *)malloc((size_t)w * h * 3);
        if (!img)
            return 2;
        so_vnoise5(img, h, w, 1234 + (uint64_t)s);
        for (int ni = 0; ni < 2; ni++) {
            size_t n = 0, nk = 0;
            so_counters ctr;
            so_timings tm;
            so_keypoint *f = so_sift(img, h, w, 3, (size_t)w * 3, 1.6, 3, ni, &n, &nk, &ctr, &tm);
            printf("%dx%d nointerval=%d: %zu features, %zu keypoints\n", w, h, ni, n, nk);
            total += n;
            if (ni == 0 && w == 64) {
                char path[1024];
                snprintf(path, sizeof path, "%s/oracle_san_features.txt", dir);
                if (so_write_features(f, n, path) != 0)
                    return 3;
            }
            if (prev && n && n_prev && n * n_prev < 4000000) {
                int *i1 = (int *)malloc(sizeof(int) * n_prev), *i2 = (int *)malloc(sizeof(int) * n_prev);
                double *best2 = (double *)malloc(sizeof(double) * 2 * n_prev);
                if (!i1 || !i2 || !best2)
                    return 2;
                so_match(prev, n_prev, f, n, 0.8, 0, i1, i2, best2);
                so_match(prev, n_prev, f, n, 0.5, 1, i1, i2, NULL);
                free(i1);
                free(i2);
                free(best2);
            }
            if (ni == 0) {
                so_free(prev);
                prev = f;
                n_prev = n;
            } else {
                so_free(f);
            }
        }
        free(img);
    }
    so_free(prev);
    printf("oracle sanitizer check ok: %zu features\n", total);
    return total > 3000 ? 0 : 1;
}
