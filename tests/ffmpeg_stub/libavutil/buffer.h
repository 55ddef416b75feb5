#pragma once
/* declaration-only subset for the lint compile of the patched vf_raisr.c (page-locked buffer pools) */
#include <stddef.h>
#include <stdint.h>
#include "hwcontext.h"
typedef struct AVBufferPool AVBufferPool;
AVBufferRef *av_buffer_create(uint8_t *data, size_t size, void (*free)(void *opaque, uint8_t *data), void *opaque, int flags);
AVBufferPool *av_buffer_pool_init2(size_t size, void *opaque, AVBufferRef *(*alloc)(void *opaque, size_t size), void (*pool_free)(void *opaque));
AVBufferRef *av_buffer_pool_get(AVBufferPool *pool);
void av_buffer_pool_uninit(AVBufferPool **pool);
