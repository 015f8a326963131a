# Build libslamhip.so (gfx950 only) and keep compiler temporaries under build/.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := slam_decomposition_amd/csrc/slam_hip.hip slam_decomposition_amd/csrc/slam_comm.hip
HDRS  := slam_decomposition_amd/csrc/slam_device.hpp slam_decomposition_amd/csrc/slam_kernels.hpp slam_decomposition_amd/csrc/slam_sampler.hpp slam_decomposition_amd/csrc/slam_weyl.hpp slam_decomposition_amd/csrc/slam_sincos.hpp slam_decomposition_amd/csrc/slam_v2.hpp slam_decomposition_amd/csrc/slam_long.hpp include/slam_hip.h
OUT   := slam_decomposition_amd/lib/libslamhip.so
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -ldl

all: $(OUT)

$(OUT): $(SRC) $(HDRS)
	mkdir -p build slam_decomposition_amd/lib
	$(HIPCC) $(FLAGS) -save-temps=obj -Rpass-analysis=kernel-resource-usage -o build/libslamhip.so $(SRC) 2> build/resource_usage.txt || (cat build/resource_usage.txt; exit 1)
	cp build/libslamhip.so $(OUT)

clean:
	rm -rf build $(OUT)

.PHONY: all clean

# host-side AddressSanitizer + UBSan build, driven through every C-ABI entry point's no-GPU paths (CPU only)
asan:
	bash tools/asan_host.sh

.PHONY: asan
