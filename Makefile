# Builds everything in-tree (the .so files travel to the GPU box with the snapshot):
#   ouster_sdk_amd/lib/libouster_hip.so        HIP kernels + C ABI (include/ouster_hip.h), gfx950
#   ouster_sdk_amd/lib/libouster_core_amd.so   C++ host API (include/ouster/core/*.h) over the C ABI
#   oracle/_build/libouster_oracle.so          CPU oracle (test infrastructure only)
HIPCC ?= hipcc
CXX ?= g++
LIB := ouster_sdk_amd/lib
CSRC := ouster_sdk_amd/csrc
HOST_SRC := $(wildcard $(CSRC)/host/*.cpp)
# EXPERIMENTS=1 also compiles the measured-slower kernel forms kept for A/B work (k_decode_wide_resolved, k_dwf_single, k_dwf_fused)
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wno-unused-result $(if $(EXPERIMENTS),-DOUSTER_EXPERIMENTS,)
# the fused decode kernels are compiled once per packet-profile specialisation (parallel with -j)
OBJ := $(CSRC)/_build
SPEC_IDS := 0 1 2 3 4 5
STREAM_IDS := 1 2 3 4 5
HIP_OBJS := $(foreach i,$(SPEC_IDS),$(OBJ)/k_decode_$(i).o) $(foreach i,$(STREAM_IDS),$(OBJ)/k_decode_stream_$(i).o) $(OBJ)/k_standalone.o $(OBJ)/ouster_hip_capi.o $(OBJ)/host_pool.o
HIP_HDRS := $(CSRC)/ouster_hip_dev.h $(CSRC)/host_pool.h $(CSRC)/kernels_common.h $(CSRC)/wide_tile.h include/ouster_hip.h
ROCM ?= /opt/rocm
CXXFLAGS := -O2 -std=c++17 -fPIC -pthread -Wall -Wextra -Iinclude -I$(CSRC)/host -I$(ROCM)/include -D__HIP_PLATFORM_AMD__

PYEXT := ouster_sdk_amd/core$(shell python3-config --extension-suffix)
PYINC := $(shell python3 -m pybind11 --includes)

all: $(LIB)/libouster_hip.so $(LIB)/libouster_core_amd.so $(PYEXT) oracle cpptests

$(OBJ)/k_decode_stream_%.o: $(CSRC)/k_decode_stream.hip $(HIP_HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DOUSTER_SPEC_ID=$* -c -o $@ $<

$(OBJ)/k_decode_%.o: $(CSRC)/k_decode.hip $(HIP_HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -DOUSTER_SPEC_ID=$* -c -o $@ $<

$(OBJ)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB)/libouster_hip.so: $(HIP_OBJS)
	mkdir -p $(LIB)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_OBJS)

$(LIB)/libouster_core_amd.so: $(HOST_SRC) $(wildcard include/ouster/core/*.h include/ouster/hip/*.h include/ouster/pcap/*.h include/ouster/osf/*.h) $(CSRC)/host/host_internal.h $(LIB)/libouster_hip.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -L$(LIB) -louster_hip -L$(ROCM)/lib -lamdhip64 -lz -l:libzstd.so.1 -Wl,-rpath,'$$ORIGIN'

$(PYEXT): $(CSRC)/python/bindings.cpp $(LIB)/libouster_core_amd.so $(wildcard include/ouster/core/*.h)
	$(CXX) -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -Iinclude $(PYINC) -o $@ $(CSRC)/python/bindings.cpp -L$(LIB) -louster_core_amd -louster_hip -Wl,-rpath,'$$ORIGIN/lib'

oracle:
	$(MAKE) -C oracle -s

cpptests: $(LIB)/libouster_core_amd.so
	$(MAKE) -C tests/cpp -s
	$(MAKE) -C oracle -s refcpptests

clean:
	rm -rf $(LIB) $(OBJ) oracle/_build tests/cpp/_build

.PHONY: all oracle clean cpptests
