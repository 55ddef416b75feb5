#pragma once
#include "avfilter.h"
AVFrame *ff_get_video_buffer(AVFilterLink *link, int w, int h);
AVFrame *ff_default_get_video_buffer(AVFilterLink *link, int w, int h);
