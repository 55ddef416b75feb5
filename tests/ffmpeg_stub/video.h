#pragma once
#include "avfilter.h"
AVFrame *ff_get_video_buffer(AVFilterLink *link, int w, int h);
