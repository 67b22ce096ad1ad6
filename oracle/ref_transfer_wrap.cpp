// ref_transfer_wrap.cpp -- C entry points over the REFERENCE's own ColorTransfer.cpp (compiled unmodified from
// /root/reference/src/common by `make -C oracle _ref` when a real <libheif/heif.h> is on the include path; that header is the
// translation unit's only external dependency, ColorTransfer.h:24-26).  Test infrastructure: tests/test_ref_pin.py diffs
// these against oracle/avif_oracle.c.  Nothing of the reference is copied here -- this file only forwards calls.
#include "ColorTransfer.h"

extern "C" {
float ref_linear_to_pq(float v, float peak) { return LinearToPQ(v, peak); }
float ref_pq_to_linear(float v, float peak) { return PQToLinear(v, peak); }
float ref_linear_to_smpte428(float v) { return LinearToSMPTE428(v); }
float ref_smpte428_to_linear(float v) { return SMPTE428ToLinear(v); }
float ref_linear_to_hlg(float v) { return LinearToHLG(v); }
float ref_hlg_to_linear(float v) { return HLGToLinear(v); }
void ref_apply_hlg_ootf(float rgb[3], const float luma[3], float gamma, float peak)
{
    const HLGLumaCoefficiants c = { luma[0], luma[1], luma[2] };
    ApplyHLGOOTF(rgb, c, gamma, peak);
}
int ref_hlg_luma_coefficients(int primaries, float out[3])
{
    try {
        const HLGLumaCoefficiants c = GetHLGLumaCoefficients(static_cast<heif_color_primaries>(primaries));
        out[0] = c.red; out[1] = c.green; out[2] = c.blue;
        return 0;
    } catch (...) { return -1; }
}
int ref_transfer_from_nclx(int tc)
{
    try { return static_cast<int>(GetTransferFunctionFromNclx(static_cast<heif_transfer_characteristics>(tc))); }
    catch (...) { return -1; }
}
}
