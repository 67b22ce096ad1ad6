// AddressSanitizer / UBSan driver for the ICC parser (csrc/icc_profile.cpp is plain host C++): mutates valid profiles the same way
// tests/test_icc_parser_fuzz.py does and calls every host entry point on HEAP copies sized exactly to the blob, so any read past the
// profile bytes is reported.  Build + run (CPU only):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I include tools/icc_fuzz_asan.cpp -o /tmp/icc_fuzz && /tmp/icc_fuzz a.icc b.icc ...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
namespace avifgpu { void set_error(const char*) {} }
#include "../avif-format_amd/csrc/icc_profile.cpp"

static uint64_t g_state = 88172645463325252ull;
static uint32_t rnd() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return (uint32_t)(g_state >> 11); }

static void call_all(const std::vector<uint8_t>& blob, bool with_clut)
{
    std::unique_ptr<uint8_t[]> exact(new uint8_t[blob.size() ? blob.size() : 1]);      // exact-size heap copy: ASan guards both ends
    if (!blob.empty()) memcpy(exact.get(), blob.data(), blob.size());
    const uint32_t n = (uint32_t)blob.size();
    avifgpu_icc_transform t;
    static avifgpu_icc_shaper8 s8;
    static avifgpu_icc_clut16 c16;
    (void)avifgpu_icc_detect(exact.get(), n);
    (void)avifgpu_icc_prepare(exact.get(), n, AVIFGPU_ICC_TARGET_REC2020_LINEAR, &t);
    (void)avifgpu_icc_prepare(exact.get(), n, AVIFGPU_ICC_TARGET_SRGB_FLOAT, &t);
    (void)avifgpu_icc_prepare_shaper8(exact.get(), n, &s8);
    if (with_clut) (void)avifgpu_icc_prepare_clut16(exact.get(), n, &c16);
}

int main(int argc, char** argv)
{
    long calls = 0;
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 1; }
        std::vector<uint8_t> icc;
        uint8_t buf[4096]; size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) icc.insert(icc.end(), buf, buf + got);
        fclose(f);
        for (size_t n = 0; n <= icc.size(); n += (n < 256 ? 1 : 29)) { call_all(std::vector<uint8_t>(icc.begin(), icc.begin() + n), false); ++calls; }
        const uint32_t count = (icc[128] << 24) | (icc[129] << 16) | (icc[130] << 8) | icc[131];
        const size_t table_end = 132 + 12 * (size_t)count;
        for (int trial = 0; trial < 3000; ++trial) {
            std::vector<uint8_t> b = icc;
            switch (trial % 6) {
            case 0: for (int k = 0; k < 1 + (int)(rnd() % 5); ++k) b[128 + rnd() % (table_end - 128)] = (uint8_t)rnd(); break;
            case 1: { const uint32_t v[5] = {0, 1, 0xffffffffu, 0x7fffffffu, count + 1000}; const uint32_t c = v[rnd() % 5];
                      b[128] = c >> 24; b[129] = c >> 16; b[130] = c >> 8; b[131] = c; break; }
            case 2: { const size_t i = 132 + 12 * (rnd() % count) + ((rnd() & 1) ? 4 : 8);
                      const uint32_t v[6] = {0xffffffffu, 0xfffffff0u, (uint32_t)icc.size() - 1, (uint32_t)icc.size(), 0x80000000u, rnd()}; const uint32_t c = v[rnd() % 6];
                      b[i] = c >> 24; b[i + 1] = c >> 16; b[i + 2] = c >> 8; b[i + 3] = c; break; }
            case 3: for (int k = 0; k < 1 + (int)(rnd() % 3); ++k) b[table_end + rnd() % (b.size() - table_end)] = (uint8_t)(rnd() % 4 == 0 ? 0xff : rnd()); break;
            case 4: { const size_t pos = table_end + rnd() % (b.size() - table_end); for (size_t i = pos; i < b.size(); ++i) b[i] = (uint8_t)rnd(); break; }
            default: b.resize(table_end + rnd() % (b.size() - table_end)); break;          // truncated inside the tag data
            }
            call_all(b, trial % 16 == 0);
            ++calls;
        }
    }
    printf("icc_fuzz_asan: %ld parser calls, no sanitizer report\n", calls);
    return 0;
}
