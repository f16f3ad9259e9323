// gsx_ops.h — C++ operator surface of the MI355X backend.  The declarations live in compat/gsplat/{Ops,Cameras,Common}.h, the
// headers a reference build includes by the reference's own file names; this header is the in-repo spelling.
#pragma once
#include "../compat/gsplat/Ops.h"
