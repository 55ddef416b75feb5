#pragma once
#define av_assert1(x) ((void)0)
