// DECLARATION-ONLY header for tests/compilecheck (see ../README.md): the FormatRecord fields and suite signatures the reference's
// headers and the adapters name.  NOT the Photoshop SDK; layout and values carry no meaning here.
#pragma once
#include "PITypes.h"
#define DLLExport
#define MACPASCAL
typedef struct PSBuffer* BufferID;
typedef OSErr (*AllocateBufferProc)(int32 size, BufferID* bufferID);
typedef Ptr (*LockBufferProc)(BufferID bufferID, Boolean moveHigh);
typedef void (*UnlockBufferProc)(BufferID bufferID);
typedef void (*FreeBufferProc)(BufferID bufferID);
typedef int32 (*BufferSpaceProc)(void);
#define kCurrentBufferProcsVersion 2
#define kCurrentBufferProcsCount 5
#define kCurrentHandleProcsVersion 1
#define kCurrentHandleProcsCount 6
struct BufferProcs { int16 bufferProcsVersion; int16 numBufferProcs; AllocateBufferProc allocateProc; LockBufferProc lockProc; UnlockBufferProc unlockProc; FreeBufferProc freeProc; BufferSpaceProc spaceProc; };
typedef Handle (*NewPIHandleProc)(int32 size);
typedef void (*DisposePIHandleProc)(Handle h);
typedef int32 (*GetPIHandleSizeProc)(Handle h);
typedef OSErr (*SetPIHandleSizeProc)(Handle h, int32 newSize);
typedef Ptr (*LockPIHandleProc)(Handle h, Boolean moveHigh);
typedef void (*UnlockPIHandleProc)(Handle h);
struct HandleProcs { int16 handleProcsVersion; int16 numHandleProcs; NewPIHandleProc newProc; DisposePIHandleProc disposeProc; GetPIHandleSizeProc getSizeProc; SetPIHandleSizeProc setSizeProc; LockPIHandleProc lockProc; UnlockPIHandleProc unlockProc; };
struct PropertyProcs;
struct PIDescriptorParameters;
enum { plugInModeGrayScale = 1, plugInModeRGBColor = 3, plugInModeGray16 = 10, plugInModeRGB48 = 11, plugInModeGray32 = 16, plugInModeRGB96 = 17 };
enum : OSErr { formatBadParameters = -30500, formatCannotRead = -30501, errPlugInHostInsufficient = -30900 };
typedef Boolean (*TestAbortProc)(void);
typedef void (*ProgressProc)(int32 done, int32 total);
typedef OSErr (*AdvanceStateProc)(void);
struct FormatRecord {
    int32 serialNumber; TestAbortProc abortProc; ProgressProc progressProc; int32 maxData; int32 minDataBytes, maxDataBytes, minRsrcBytes, maxRsrcBytes;
    intptr_t dataFork, rsrcFork; void* fileSpec; int16 imageMode; Point imageSize; int16 depth; int16 planes; double imageHRes, imageVRes;
    void* redLUT; void* greenLUT; void* blueLUT; void* data; Rect theRect; int16 loPlane, hiPlane, colBytes; int32 rowBytes, planeBytes;
    int16 planeMap[24]; Boolean canTranspose, needTranspose; OSType hostSig; void* hostProc; int16 hostModes; Handle revertInfo;
    void* hostNewHdl; void* hostDisposeHdl; Handle imageRsrcData; int32 imageRsrcSize; void* plugInMonitor; void* platformData;
    BufferProcs* bufferProcs; void* resourceProcs; void* processEvent; void* displayPixels; HandleProcs* handleProcs;
    OSType fileType; void* colorServices; AdvanceStateProc advanceState; PropertyProcs* propertyProcs; void* imageServicesProcs;
    int16 tileWidth, tileHeight; Point tileOrigin; PIDescriptorParameters* descriptorParameters; void* errorString; void* channelPortProcs;
    void* documentInfo; void* sSPBasic; void* plugInRef; int32 transparentIndex; Handle iCCprofileData; int32 iCCprofileSize; int32 canUseICCProfiles;
    int32 lutCount; int32 preferredColorModes; int32 convertMode; OSErr dataForkOrRsrcForkResult; int32 layerData; void* layerName; void* pluginUsing32BitCoordinates;
    int32 HostSupports32BitCoordinates; int32 PluginUsing32BitCoordinates; VPoint imageSize32; VRect theRect32; VPoint tileOrigin32;
    int32 transparencyPlane; int32 transparencyMatting; Boolean premultipliedAlpha; int32 maxValue; void* metaDataProcs; int32 hostInSecondaryProcess;
};
typedef FormatRecord* FormatRecordPtr;
