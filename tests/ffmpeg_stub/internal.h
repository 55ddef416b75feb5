#pragma once
#include "avfilter.h"
int ff_filter_frame(AVFilterLink *link, AVFrame *frame);
int ff_request_frame(AVFilterLink *link);
