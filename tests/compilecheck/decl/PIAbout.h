// DECLARATION-ONLY header for tests/compilecheck (see ../README.md).  NOT the Photoshop SDK.
#pragma once
struct AboutRecord;
typedef AboutRecord* AboutRecordPtr;
