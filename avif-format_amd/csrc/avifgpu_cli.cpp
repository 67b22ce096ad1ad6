// avifgpu_cli -- raw <-> planes converter that drives the FormatRecord tile protocol of include/avifgpu_host.h without
// Photoshop.  The CLI plays the host: advanceState() feeds rows of a raw file (save direction) or
// collects them (open direction), exactly what Photoshop does for the plug-in's row loops (WriteHeifImage.cpp:1017-1029,
// ReadHeifImage.cpp:141-160).  No conversion happens here: every pixel goes through libavifgpu (MI355X or error).
//
//   raw file    : host rows, tightly packed, Photoshop layout (interleaved planes, 8 / 16 [0..32768] / 32-bit float)
//   planes file : the heif_image planes in order 0..3, each with tight rows (width * bytes-per-sample), no header
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/avifgpu_host.h"

namespace {

struct Host {                       // the callbacks carry no user pointer in the SDK either: one static host
    avifgpu_FormatRecord fr{};
    FILE* file = nullptr;
    bool saving = false;            // true: host -> plug-in
    int tiles = 0;
} g_host;

avifgpu_OSErr advance_state()
{
    const avifgpu_VRect& r = g_host.fr.theRect32;
    const size_t bytes = (size_t)(r.bottom - r.top) * (size_t)g_host.fr.rowBytes;
    ++g_host.tiles;
    if (fseeko(g_host.file, (off_t)r.top * g_host.fr.rowBytes, SEEK_SET) != 0) return AVIFGPU_readErr;
    if (g_host.saving) return fread(g_host.fr.data, 1, bytes, g_host.file) == bytes ? AVIFGPU_noErr : AVIFGPU_readErr;
    return fwrite(g_host.fr.data, 1, bytes, g_host.file) == bytes ? AVIFGPU_noErr : AVIFGPU_writErr;
}
uint8_t never_abort() { return 0; }

struct Args {
    std::string cmd, in, out, icc;
    int width = 0, height = 0, depth = 8, planes = 3, bits = 8, transfer = AVIFGPU_TRANSFER_CLIP, peak = 1000;
    int alpha = AVIFGPU_ALPHA_NONE, output = AVIFGPU_OUT_REFERENCE, chroma = AVIFGPU_CHROMA_444;
    int matrix = AVIFGPU_MATRIX_BT601, primaries = AVIFGPU_PRIMARIES_BT709, tc = 2, limited = 0, colorspace = AVIFGPU_COLORSPACE_YCBCR;
    int lossless = 0, maxdata = 0, device = 0, hlg_ootf = 0, nclx = 0, keep_profile = 0;
    float gamma = 1.2f;
};

[[noreturn]] void usage(const char* why)
{
    if (why) fprintf(stderr, "avifgpu_cli: %s\n", why);
    fprintf(stderr,
        "usage: avifgpu_cli write --width W --height H --depth 8|16|32 --planes 1..4 --bits 8|10|12 [--transfer clip|pq|smpte428]\n"
        "                         [--peak NITS] [--alpha none|straight|premultiplied] [--ycbcr 444|422|420] [--matrix N] [--primaries N]\n"
        "                         [--lossless] [--icc PROFILE [--keep-profile]] [--maxdata BYTES] [--device N] IN.raw OUT.planes\n"
        "       avifgpu_cli read  --width W --height H --depth 8|16|32 --bits 8|10|12 --colorspace ycbcr|rgb|mono [--chroma 444|422|420]\n"
        "                         [--alpha none|straight|premultiplied] [--matrix N --primaries N --tc N [--limited]] [--peak NITS]\n"
        "                         [--hlg-ootf --gamma G] [--maxdata BYTES] [--device N] IN.planes OUT.raw\n");
    exit(2);
}

int pick(const char* v, std::initializer_list<std::pair<const char*, int>> tbl, const char* opt)
{
    for (auto& e : tbl) if (!strcmp(v, e.first)) return e.second;
    usage((std::string("bad value for ") + opt + ": " + v).c_str());
}

Args parse(int argc, char** argv)
{
    Args a;
    if (argc < 2) usage(nullptr);
    a.cmd = argv[1];
    if (a.cmd != "write" && a.cmd != "read") usage("first argument must be write or read");
    std::vector<std::string> pos;
    for (int i = 2; i < argc; ++i) {
        const std::string o = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) usage(("missing value for " + o).c_str()); return argv[++i]; };
        if (o == "--width") a.width = atoi(val());
        else if (o == "--height") a.height = atoi(val());
        else if (o == "--depth") a.depth = atoi(val());
        else if (o == "--planes") a.planes = atoi(val());
        else if (o == "--bits") a.bits = atoi(val());
        else if (o == "--peak") a.peak = atoi(val());
        else if (o == "--matrix") { a.matrix = atoi(val()); a.nclx = 1; }
        else if (o == "--primaries") { a.primaries = atoi(val()); a.nclx = 1; }
        else if (o == "--tc") { a.tc = atoi(val()); a.nclx = 1; }
        else if (o == "--limited") { a.limited = 1; a.nclx = 1; }
        else if (o == "--maxdata") a.maxdata = atoi(val());
        else if (o == "--device") a.device = atoi(val());
        else if (o == "--gamma") a.gamma = (float)atof(val());
        else if (o == "--hlg-ootf") a.hlg_ootf = 1;
        else if (o == "--lossless") a.lossless = 1;
        else if (o == "--keep-profile") a.keep_profile = 1;
        else if (o == "--icc") a.icc = val();
        else if (o == "--transfer") a.transfer = pick(val(), {{"clip", AVIFGPU_TRANSFER_CLIP}, {"pq", AVIFGPU_TRANSFER_PQ}, {"smpte428", AVIFGPU_TRANSFER_SMPTE428}}, "--transfer");
        else if (o == "--alpha") a.alpha = pick(val(), {{"none", AVIFGPU_ALPHA_NONE}, {"straight", AVIFGPU_ALPHA_STRAIGHT}, {"premultiplied", AVIFGPU_ALPHA_PREMULTIPLIED}}, "--alpha");
        else if (o == "--ycbcr") { a.output = AVIFGPU_OUT_YCBCR; a.chroma = pick(val(), {{"444", AVIFGPU_CHROMA_444}, {"422", AVIFGPU_CHROMA_422}, {"420", AVIFGPU_CHROMA_420}}, "--ycbcr"); }
        else if (o == "--chroma") a.chroma = pick(val(), {{"444", AVIFGPU_CHROMA_444}, {"422", AVIFGPU_CHROMA_422}, {"420", AVIFGPU_CHROMA_420}}, "--chroma");
        else if (o == "--colorspace") a.colorspace = pick(val(), {{"ycbcr", AVIFGPU_COLORSPACE_YCBCR}, {"rgb", AVIFGPU_COLORSPACE_RGB}, {"mono", AVIFGPU_COLORSPACE_MONOCHROME}}, "--colorspace");
        else if (o == "--help" || o == "-h") usage(nullptr);
        else if (o.rfind("--", 0) == 0) usage(("unknown option " + o).c_str());
        else pos.push_back(o);
    }
    if (pos.size() != 2) usage("need exactly IN and OUT files");
    a.in = pos[0]; a.out = pos[1];
    if (a.width <= 0 || a.height <= 0) usage("--width and --height are required");
    return a;
}

int image_mode(int depth, bool mono)
{
    if (mono) return depth == 8 ? avifgpu_plugInModeGrayScale : depth == 16 ? avifgpu_plugInModeGray16 : avifgpu_plugInModeGray32;
    return depth == 8 ? avifgpu_plugInModeRGBColor : depth == 16 ? avifgpu_plugInModeRGB48 : avifgpu_plugInModeRGB96;
}

void setup_record(const Args& a, int planes)
{
    avifgpu_FormatRecord& fr = g_host.fr;
    fr.abortProc = never_abort; fr.advanceState = advance_state; fr.progressProc = nullptr;
    fr.maxData = a.maxdata;
    fr.depth = (int16_t)a.depth; fr.planes = (int16_t)planes;
    fr.imageMode = (int16_t)image_mode(a.depth, planes <= 2);
    fr.imageSize32 = {a.height, a.width};
    fr.imageSize = {(int16_t)(a.height > 32767 ? 32767 : a.height), (int16_t)(a.width > 32767 ? 32767 : a.width)};
    fr.HostSupports32BitCoordinates = 1; fr.PluginUsing32BitCoordinates = 1;
}

int plane_rows(const avifgpu_image& img, int pl)
{
    const bool sub = img.colorspace == AVIFGPU_COLORSPACE_YCBCR && (pl == 1 || pl == 2) && img.chroma == AVIFGPU_CHROMA_420;
    return sub ? (img.height + 1) >> 1 : img.height;
}
size_t plane_row_bytes(const avifgpu_image& img, int pl)
{
    const int ssz = img.bit_depth > 8 ? 2 : 1;
    if (img.colorspace == AVIFGPU_COLORSPACE_RGB && img.chroma >= 10)
        return (size_t)img.width * ((img.chroma == 11 || img.chroma == 15) ? 4 : 3) * ssz;
    const bool sub = img.colorspace == AVIFGPU_COLORSPACE_YCBCR && (pl == 1 || pl == 2) && img.chroma != AVIFGPU_CHROMA_444;
    return (size_t)(sub ? (img.width + 1) >> 1 : img.width) * ssz;
}

int fail(const char* what, int code)
{
    fprintf(stderr, "avifgpu_cli: %s failed: OSErr %d (%s)\n", what, code, avifgpu_last_error());
    return 1;
}

int do_write(const Args& a)
{
    setup_record(a, a.planes);
    std::vector<uint8_t> profile;
    if (!a.icc.empty()) {
        FILE* f = fopen(a.icc.c_str(), "rb");
        if (!f) { perror(a.icc.c_str()); return 1; }
        uint8_t buf[65536]; size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) profile.insert(profile.end(), buf, buf + n);
        fclose(f);
        g_host.fr.iCCprofileData = profile.data(); g_host.fr.iCCprofileSize = (int32_t)profile.size();
    }
    g_host.file = fopen(a.in.c_str(), "rb");
    if (!g_host.file) { perror(a.in.c_str()); return 1; }
    g_host.saving = true;
    avifgpu_SaveUIOptions o{};
    o.imageBitDepth = a.bits; o.hdrTransferFunction = a.transfer; o.pq.nominalPeakBrightness = a.peak;
    o.chromaSubsampling = a.chroma; o.lossless = (uint8_t)a.lossless;
    o.keepColorProfile = (uint8_t)a.keep_profile;
    o.iccDecision = AVIFGPU_ICC_LIKE_PLUGIN;
    if (!a.icc.empty()) {
        // the plug-in's own gate (ColorProfileConversion.cpp:98-157), taken inside the library: HDR saves convert to Rec.2020
        // unless the document already is Rec.2020; 8/16-bit saves convert to sRGB unless it already is sRGB or the profile is
        // kept; 32-bit Clip saves convert to sRGB unless the profile is kept.  Printed here, decided there.
        const int32_t conversion = avifgpu_host_required_conversion_for_record(&g_host.fr, &o);
        if (conversion < 0) return fail("avifgpu_host_required_conversion_for_record", conversion);
        fprintf(stderr, "icc: %s\n", conversion == AVIFGPU_CONVERT_TO_REC2020 ? "convert to Rec.2020"
                                    : conversion == AVIFGPU_CONVERT_TO_SRGB ? "convert to sRGB" : "no conversion");
    }
    avifgpu_image img{};
    const int rc = avifgpu_host_create_heif_image(&g_host.fr, a.alpha, &o, a.output, a.lossless ? AVIFGPU_MATRIX_RGB_GBR : a.matrix,
                                                  a.primaries, &img);
    fclose(g_host.file);
    if (rc) return fail("avifgpu_host_create_heif_image", rc);
    FILE* out = fopen(a.out.c_str(), "wb");
    if (!out) { perror(a.out.c_str()); return 1; }
    size_t total = 0;
    for (int pl = 0; pl < 4; ++pl) {
        if (!img.plane[pl]) continue;
        const size_t rb = plane_row_bytes(img, pl);
        for (int y = 0; y < plane_rows(img, pl); ++y) total += fwrite(img.plane[pl] + (size_t)y * img.stride[pl], 1, rb, out);
    }
    fclose(out);
    fprintf(stderr, "write: %dx%d depth %d -> %d-bit %s, %d tiles, %zu bytes\n", a.width, a.height, a.depth, a.bits,
            img.colorspace == AVIFGPU_COLORSPACE_YCBCR ? "YCbCr planes" : img.colorspace == AVIFGPU_COLORSPACE_RGB ? "interleaved RGB" : "mono",
            g_host.tiles, total);
    avifgpu_image_free(&img);
    return 0;
}

int do_read(const Args& a)
{
    avifgpu_image img{};
    img.width = a.width; img.height = a.height; img.colorspace = a.colorspace; img.bit_depth = a.bits;
    img.chroma = a.colorspace == AVIFGPU_COLORSPACE_MONOCHROME ? AVIFGPU_CHROMA_MONOCHROME
               : a.colorspace == AVIFGPU_COLORSPACE_RGB ? AVIFGPU_CHROMA_444 : a.chroma;
    img.has_alpha = a.alpha != AVIFGPU_ALPHA_NONE; img.premultiplied_alpha = a.alpha == AVIFGPU_ALPHA_PREMULTIPLIED;
    if (avifgpu_image_alloc(&img) != AVIFGPU_noErr) { fprintf(stderr, "avifgpu_cli: out of memory\n"); return 1; }
    FILE* in = fopen(a.in.c_str(), "rb");
    if (!in) { perror(a.in.c_str()); return 1; }
    for (int pl = 0; pl < 4; ++pl) {
        if (!img.plane[pl]) continue;
        const size_t rb = plane_row_bytes(img, pl);
        for (int y = 0; y < plane_rows(img, pl); ++y)
            if (fread(img.plane[pl] + (size_t)y * img.stride[pl], 1, rb, in) != rb) { fprintf(stderr, "avifgpu_cli: %s is too short\n", a.in.c_str()); return 1; }
    }
    fclose(in);
    const int planes = (a.colorspace == AVIFGPU_COLORSPACE_MONOCHROME ? 1 : 3) + (img.has_alpha ? 1 : 0);
    setup_record(a, planes);
    g_host.file = fopen(a.out.c_str(), "wb");
    if (!g_host.file) { perror(a.out.c_str()); return 1; }
    g_host.saving = false;
    avifgpu_nclx nclx{a.primaries, a.tc, a.matrix, (uint8_t)!a.limited};
    avifgpu_LoadUIOptions lo{};
    lo.pq.nominalPeakBrightness = a.peak; lo.hlg.applyOOTF = (uint8_t)a.hlg_ootf; lo.hlg.displayGamma = a.gamma; lo.hlg.nominalPeakBrightness = a.peak;
    const int rc = avifgpu_host_read_heif_image(&img, a.alpha, (a.nclx || a.depth == 32) ? &nclx : nullptr, &lo, &g_host.fr);
    fclose(g_host.file);
    avifgpu_image_free(&img);
    if (rc) return fail("avifgpu_host_read_heif_image", rc);
    fprintf(stderr, "read: %dx%d %d-bit -> host depth %d, %d planes, %d tiles, maxValue %d\n", a.width, a.height, a.bits, a.depth,
            planes, g_host.tiles, g_host.fr.maxValue);
    return 0;
}

} // namespace

int main(int argc, char** argv)
{
    const Args a = parse(argc, argv);
    const int rc = avifgpu_init(a.device);
    if (rc) return fail("avifgpu_init", rc);
    const int r = a.cmd == "write" ? do_write(a) : do_read(a);
    avifgpu_shutdown();
    return r;
}
