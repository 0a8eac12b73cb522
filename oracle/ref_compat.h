/*
 * ref_compat.h — force-included (-include) when compiling the UNMODIFIED reference sources
 * under /root/reference with g++.  Test infrastructure only (oracle/_ref build).
 *
 * The reference was written for MSVC: it relies on transitive standard includes and on
 * std::cosf/sinf/powf/fmaxf/fmodf, which libstdc++ 11 does not declare in namespace std
 * (the functions themselves are the ordinary libm ones).  Nothing here replaces a header,
 * library or tool the image lacks: it only pulls in standard headers and re-exports the
 * global libm float functions into namespace std.
 */
#pragma once
#include <cmath>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
namespace std {
using ::cosf;
using ::fmaxf;
using ::fmodf;
using ::powf;
using ::sinf;
}  // namespace std
