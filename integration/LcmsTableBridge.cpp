// integration/LcmsTableBridge.cpp -- see LcmsTableBridge.h.  Little CMS public API only.
#include "LcmsTableBridge.h"

#include <lcms2.h>

#include <memory>

namespace
{
    struct ContextDeleter { using pointer = cmsContext; void operator()(cmsContext c) const noexcept { if (c) cmsDeleteContext(c); } };
    struct ProfileDeleter { void operator()(cmsHPROFILE p) const noexcept { if (p) cmsCloseProfile(p); } };
    struct TransformDeleter { void operator()(cmsHTRANSFORM t) const noexcept { if (t) cmsDeleteTransform(t); } };
    using ScopedContext = std::unique_ptr<_cmsContext_struct, ContextDeleter>;        // the reference keeps the same three RAII handles
    using ScopedProfile = std::unique_ptr<void, ProfileDeleter>;          // (ScopedLcms.h)
    using ScopedTransform = std::unique_ptr<void, TransformDeleter>;

    struct TwoTransforms { cmsHTRANSFORM words, floats; };

    void RunWords(void* user, const uint16_t* in, uint16_t* out, uint32_t pixels)
    {
        cmsDoTransform(static_cast<TwoTransforms*>(user)->words, in, out, pixels);
    }

    void RunBytes(void* user, const uint8_t* in, uint8_t* out, uint32_t pixels)
    {
        cmsDoTransform(static_cast<TwoTransforms*>(user)->words, in, out, pixels);
    }

    void RunFloats(void* user, const float* in, float* out, uint32_t pixels)
    {
        cmsDoTransform(static_cast<TwoTransforms*>(user)->floats, in, out, pixels);
    }
}

extern "C" int32_t avifgpu_lcms_document_to_srgb_clut16(const void* iccProfile, uint32_t size, avifgpu_icc_clut16* out)
{
    if (!iccProfile || size == 0 || !out) return AVIFGPU_formatBadParameters;

    ScopedContext context(cmsCreateContext(nullptr, nullptr));
    if (!context) return AVIFGPU_formatCannotRead;
    ScopedProfile document(cmsOpenProfileFromMemTHR(context.get(), iccProfile, size));
    ScopedProfile srgb(cmsCreate_sRGBProfileTHR(context.get()));
    if (!document || !srgb || cmsGetColorSpace(document.get()) != cmsSigRgbData) return AVIFGPU_formatCannotRead;

    const cmsUInt32Number flags = cmsFLAGS_BLACKPOINTCOMPENSATION;        // ColorProfileConversion.cpp:278
    ScopedTransform words(cmsCreateTransformTHR(context.get(), document.get(), TYPE_RGB_16, srgb.get(), TYPE_RGB_16, INTENT_PERCEPTUAL, flags));
    ScopedTransform floats(cmsCreateTransformTHR(context.get(), document.get(), TYPE_RGB_FLT, srgb.get(), TYPE_RGB_FLT, INTENT_PERCEPTUAL, flags));
    if (!words || !floats) return AVIFGPU_formatCannotRead;

    TwoTransforms both{ words.get(), floats.get() };
    return avifgpu_icc_clut16_from_transforms(RunFloats, RunWords, &both, out);
}

// Round 6: the same for an 8-bit document -- the TYPE_RGB_8 transform InitializeForSRGBConversion creates for hostBitsPerChannel == 8
// (ColorProfileConversion.cpp:280-289) in the proof's slot.  A matrix/TRC profile is refused by the proof (lcms2 runs its matrix-shaper
// there): the library's own avifgpu_icc_prepare_shaper8 has taken it before an adapter gets here.
extern "C" int32_t avifgpu_lcms_document_to_srgb_clut8(const void* iccProfile, uint32_t size, avifgpu_icc_clut16* out)
{
    if (!iccProfile || size == 0 || !out) return AVIFGPU_formatBadParameters;

    ScopedContext context(cmsCreateContext(nullptr, nullptr));
    if (!context) return AVIFGPU_formatCannotRead;
    ScopedProfile document(cmsOpenProfileFromMemTHR(context.get(), iccProfile, size));
    ScopedProfile srgb(cmsCreate_sRGBProfileTHR(context.get()));
    if (!document || !srgb || cmsGetColorSpace(document.get()) != cmsSigRgbData) return AVIFGPU_formatCannotRead;

    const cmsUInt32Number flags = cmsFLAGS_BLACKPOINTCOMPENSATION;        // ColorProfileConversion.cpp:278
    ScopedTransform bytes(cmsCreateTransformTHR(context.get(), document.get(), TYPE_RGB_8, srgb.get(), TYPE_RGB_8, INTENT_PERCEPTUAL, flags));
    ScopedTransform floats(cmsCreateTransformTHR(context.get(), document.get(), TYPE_RGB_FLT, srgb.get(), TYPE_RGB_FLT, INTENT_PERCEPTUAL, flags));
    if (!bytes || !floats) return AVIFGPU_formatCannotRead;

    TwoTransforms both{ bytes.get(), floats.get() };
    return avifgpu_icc_clut8_from_transforms(RunFloats, RunBytes, &both, out);
}
