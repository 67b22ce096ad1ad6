// DECLARATION-ONLY header for tests/compilecheck (see ../README.md): the scalar types and error codes the reference's headers name
// (SURVEY.md Appendix A).  NOT the Photoshop SDK; nothing is ever compiled to an object or linked against it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <math.h>
typedef int8_t int8; typedef int16_t int16; typedef int32_t int32; typedef int64_t int64;
typedef uint8_t uint8; typedef uint16_t uint16; typedef uint32_t uint32; typedef uint64_t unsigned64;
typedef double real64;
typedef unsigned char Boolean;
typedef int16 OSErr;
typedef char* Ptr;
typedef Ptr* Handle;
typedef uint32 OSType;
typedef uint32 ResType;
struct VPoint { int32 v, h; };
struct VRect { int32 top, left, bottom, right; };
struct Point { int16 v, h; };
struct Rect { int16 top, left, bottom, right; };
enum : OSErr { noErr = 0, userCanceledErr = -128, readErr = -19, writErr = -20, eofErr = -39, memFullErr = -108, nilHandleErr = -109, paramErr = -50 };
#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
