// radar.h -- umbrella header (src/radar.h:15-18) for the detect + locate path.
#pragma once
#include "detector.h"
#include "locator.h"
#include "robot.h"
