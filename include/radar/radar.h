// radar.h -- umbrella header (src/radar.h:15-18): detect + locate, and the tracker stage after them.
#pragma once
#include "detector.h"
#include "locator.h"
#include "robot.h"
#include "tracker.h"
