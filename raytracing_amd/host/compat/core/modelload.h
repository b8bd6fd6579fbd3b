// core/modelload.h — forwarding header: the reference's scene scripts include "core/modelload.h"
// (RayTracing/RayTracing.cpp:4-21); every class they use from it is declared in rt_host.h.
#pragma once
#include "../../rt_host.h"
