// rtweekend.h — forwarding header (RayTracing/RayTracing.cpp:4 includes it; nothing from it is used by the live scenes).
#pragma once
#include "../rt_host.h"
