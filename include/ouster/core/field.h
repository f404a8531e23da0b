// field.h -- Field / FieldType (ouster_core/include/ouster/core/field.h): in this mirror they live in lidar_frame.h.
#pragma once
#include "ouster/core/lidar_frame.h"
