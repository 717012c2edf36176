// Shared by the translation units of librtx: the thread-local error text behind rtx_last_error().
#pragma once
#include <cstdarg>

#if defined(__GNUC__)
#define RTX_PRINTF_LIKE __attribute__((format(printf, 1, 2)))
#else
#define RTX_PRINTF_LIKE
#endif

// Records a printf-formatted message for rtx_last_error() on the calling thread.
void rtx_set_error(const char* fmt, ...) RTX_PRINTF_LIKE;
