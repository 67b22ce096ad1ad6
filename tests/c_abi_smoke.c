/* A plain C99 client of the drop-in boundary: what a plug-in written in C (or any FFI) does.  Converts a 5x3 8-bit RGBA
 * document to the reference hand-off (interleaved 8-bit RGBA, premultiplied) with host pointers and checks three pixels by hand:
 * PremultiplyColor(c, a, 255) = min(roundf(c * a / 255), 255) (reference PremultipliedAlpha.cpp:54-61).
 * Exit codes: 0 ok, 3 no HIP device (the library has no CPU fallback), 1 anything else. */
#include <stdio.h>
#include <string.h>
#include "avifgpu.h"

int main(void)
{
    int rc = avifgpu_init(0);
    if (rc != 0) { fprintf(stderr, "avifgpu_init: %d (%s)\n", rc, avifgpu_last_error()); return 3; }
    enum { W = 5, H = 3 };
    unsigned char src[H][W * 4], dst[H][W * 4];
    int x, y;
    for (y = 0; y < H; ++y)
        for (x = 0; x < W; ++x) {
            src[y][4 * x + 0] = (unsigned char)(40 * x + y);
            src[y][4 * x + 1] = 200;
            src[y][4 * x + 2] = 255;
            src[y][4 * x + 3] = (unsigned char)(x == 0 ? 0 : x == 4 ? 255 : 64 * x);
        }
    avifgpu_write_desc d;
    memset(&d, 0, sizeof d);
    d.width = W; d.height = H; d.depth = 8; d.planes = 4; d.bit_depth = 8;
    d.alpha_state = AVIFGPU_ALPHA_PREMULTIPLIED; d.output = AVIFGPU_OUT_REFERENCE; d.full_range = 1;
    void* planes[4] = { dst, 0, 0, 0 };
    int64_t strides[4] = { W * 4, 0, 0, 0 };
    rc = avifgpu_write_rows(&d, 0, H, src, W * 4, planes, strides, AVIFGPU_MEM_HOST, 0);
    if (rc != 0) { fprintf(stderr, "avifgpu_write_rows: %d (%s)\n", rc, avifgpu_last_error()); return 1; }
    /* x = 0: alpha 0 -> colour 0;  x = 4: alpha 255 -> unchanged;  x = 2: alpha 128: 200*128/255 = 100.39 -> 100, 255 -> 128 */
    if (dst[1][0] != 0 || dst[1][1] != 0 || dst[1][2] != 0 || dst[1][3] != 0) return 1;
    if (dst[1][16] != 161 || dst[1][17] != 200 || dst[1][18] != 255 || dst[1][19] != 255) return 1;
    if (dst[1][8] != 41 /* 81*128/255 = 40.66 */ || dst[1][9] != 100 || dst[1][10] != 128 || dst[1][11] != 128) return 1;
    printf("c_abi_smoke ok (%s)\n", avifgpu_last_kernel_name());
    avifgpu_shutdown();
    return 0;
}
