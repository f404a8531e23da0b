// impl/cartesian.h -- impl::cartesianT<T> (ouster_core/include/ouster/core/impl/cartesian.h:36-105): in this mirror it is
// declared with the lookup tables in xyzlut.h.
#pragma once
#include "ouster/core/xyzlut.h"
