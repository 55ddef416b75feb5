#pragma once
#include <stddef.h>
#include <stdint.h>
typedef struct AVDRMObjectDescriptor { int fd; size_t size; uint64_t format_modifier; } AVDRMObjectDescriptor;
typedef struct AVDRMPlaneDescriptor { int object_index; ptrdiff_t offset; ptrdiff_t pitch; } AVDRMPlaneDescriptor;
typedef struct AVDRMLayerDescriptor { uint32_t format; int nb_planes; AVDRMPlaneDescriptor planes[4]; } AVDRMLayerDescriptor;
typedef struct AVDRMFrameDescriptor { int nb_objects; AVDRMObjectDescriptor objects[4]; int nb_layers; AVDRMLayerDescriptor layers[4]; } AVDRMFrameDescriptor;
