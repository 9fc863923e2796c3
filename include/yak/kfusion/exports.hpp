// yak/kfusion/exports.hpp -- symbol visibility (the reference header of the same path carries the
// Windows dllexport dance; only the macro name is API).
#pragma once
#if defined(__GNUC__)
#define KF_EXPORTS __attribute__((visibility("default")))
#else
#define KF_EXPORTS
#endif
