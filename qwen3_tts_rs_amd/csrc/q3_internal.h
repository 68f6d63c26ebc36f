// q3_internal.h — shared between the translation units of libq3tts.so (not part of the C ABI).
#pragma once
#include "../../include/q3tts.h"

// sets the thread-local message q3_last_error() returns and hands back `st` (defined in q3_model.hip)
extern "C" q3_status q3i_set_err(q3_status st, const char* fmt, ...) __attribute__((format(printf, 2, 3), visibility("hidden")));

#define Q3I_CHECK(expr)                   \
    do {                                  \
        q3_status s_ = (expr);            \
        if (s_ != Q3_OK) return s_;       \
    } while (0)
