// f360/parameters.h -- the geometry constants of the reference's src/parameters.h:8-9 and the
// reduced-size rule of src/run_satlogrectilinear.cc:368-369.
#pragma once
#include <cmath>

#define REDUCED_BUFFER_WIDTH 1072
#define REDUCED_BUFFER_HEIGHT 608

inline int f360_reduced_size(int full) { return 16 * (int)std::ceil(full / 1.8 / 16); }
