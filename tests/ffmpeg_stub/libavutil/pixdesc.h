#pragma once
#include "pixfmt.h"
