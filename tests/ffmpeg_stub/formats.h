#pragma once
#include "avfilter.h"
AVFilterFormats *ff_make_format_list(const int *fmts);
int ff_set_common_formats(AVFilterContext *ctx, AVFilterFormats *f);
