// TEST INFRASTRUCTURE — instantiates the reference's vendored tinyexr (dep/tinyexr.h) for oracle/_ref/adypt_ref.
// In the reference this happens inside src/Tracer/OglPathTracer.cpp:8-9, which also needs a live GL context.
#define TINYEXR_IMPLEMENTATION
#include <tinyexr.h>
