#pragma once
// TEST INFRASTRUCTURE (oracle/): Realtime Math stand-in. mask helpers live in vector4f.h.
#include "rtm/vector4f.h"
