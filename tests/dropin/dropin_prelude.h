// Force-included in front of the reference's entry file for the drop-in check: system headers first, then rename the two
// symbols of that file that collide on Linux (its main(), and its `static Vector3f random()` vs glibc's random()).
#include <cstdlib>
#include <cmath>
#include <iostream>
#include <fstream>
#include <time.h>
#include "rt_host.h"
#define main reference_main
#define random ref_random
