// DECLARATION-ONLY header for tests/compilecheck (see ../README.md): NOT the Photoshop SDK, never used to build anything.
#pragma once
#define __PIWin__ 0
