// internal: pulls in the public C ABI (include/msc.h) for the kernel translation units
#pragma once
#include "../../include/msc.h"
