#pragma once
#include <stddef.h>
#include <limits.h>
enum AVOptionType { AV_OPT_TYPE_FLOAT, AV_OPT_TYPE_INT, AV_OPT_TYPE_STRING };
#define AV_OPT_FLAG_FILTERING_PARAM 1
#define AV_OPT_FLAG_VIDEO_PARAM 2
typedef struct AVOption { const char *name, *help; int offset; enum AVOptionType type;
    union { long long i64; double dbl; const char *str; } default_val; double min, max; int flags; } AVOption;
typedef struct AVClass { const char *class_name; const AVOption *option; } AVClass;
