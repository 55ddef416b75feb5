#pragma once
int av_image_get_linesize(int pix_fmt, int width, int plane);
#include <stdint.h>
int av_image_fill_linesizes(int linesizes[4], int pix_fmt, int width);
int av_image_fill_pointers(uint8_t *data[4], int pix_fmt, int height, uint8_t *ptr, const int linesizes[4]);
#ifndef FFALIGN
#define FFALIGN(x, a) (((x) + (a) - 1) & ~((a) - 1))
#endif
