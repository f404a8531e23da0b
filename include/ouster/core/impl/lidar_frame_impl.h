// impl/lidar_frame_impl.h -- visit_field / foreach_channel_field / frame_to_packets / destagger_into of the reference
// (ouster_core/include/ouster/core/impl/lidar_frame_impl.h): in this mirror they are declared in lidar_frame.h.
#pragma once
#include "ouster/core/lidar_frame.h"
