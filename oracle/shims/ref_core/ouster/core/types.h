// TEST INFRASTRUCTURE (oracle/): see lidar_frame.h next to this file.
#pragma once
#include "ouster/core/lidar_frame.h"
