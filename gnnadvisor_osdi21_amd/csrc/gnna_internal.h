// gnna_internal.h -- shared by the translation units of libgnna.so (not installed).
#ifndef GNNA_INTERNAL_H_
#define GNNA_INTERNAL_H_

#include "gnna.h"

namespace gnna {
// Records a formatted message for gnna_last_error() on this thread and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace gnna

#endif
