// os_pcap.h -- the reference keeps its capture playback / recording helpers here (ouster_pcap/include/ouster/pcap/os_pcap.h);
// of those this mirror has the reader only (pcap.h).  The header exists so that code including it compiles unchanged; like
// the reference's it brings the core types along.
#pragma once
#include "ouster/core/lidar_frame.h"
#include "ouster/core/types.h"
#include "ouster/pcap/pcap.h"
