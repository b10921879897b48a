// hast_internal.h -- what the translation units of libhast.so share besides the public ABI.
#pragma once
#include "../../include/hast.h"

namespace hast {
hast_status set_error(hast_status st, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   // text for hast_last_error()
int default_minimizer_for(int k);                                                                    // honours HAST_MINIMIZER
}  // namespace hast
