#pragma once
int av_image_get_linesize(int pix_fmt, int width, int plane);
