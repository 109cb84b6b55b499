# Convenience targets for C/C++ consumers; `python -c "import __graft_entry__ as g; g.build()"` does the same build.
HIPCC ?= /opt/rocm/bin/hipcc
CXX ?= g++
LIB = splatapult_amd/lib/libmsplat.so
SRC = splatapult_amd/csrc/msplat_device.hip splatapult_amd/csrc/msplat_group.hip splatapult_amd/host/gaussian_scene.cpp splatapult_amd/host/scene_config.cpp \
      splatapult_amd/host/point_scene.cpp
HDR = $(wildcard splatapult_amd/csrc/*.hip.h) $(wildcard splatapult_amd/csrc/*.hip.inc) splatapult_amd/host/gaussian_scene.hpp splatapult_amd/host/scene_config.hpp \
      splatapult_amd/host/point_scene.hpp include/msplat.h include/msplat_debug.h

all: $(LIB) examples

$(LIB): $(SRC) $(HDR)
	mkdir -p splatapult_amd/lib
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function -o $@ $(SRC) -lpthread

examples: build/example_render build/example_points

build/example_%: splatapult_amd/host/example_%.cpp $(LIB) splatapult_amd/host/msplat_host.hpp
	mkdir -p build
	$(CXX) -std=c++17 -I. -Isplatapult_amd/host $< -Lsplatapult_amd/lib -lmsplat -Wl,-rpath,$(CURDIR)/splatapult_amd/lib -o $@

oracle:
	$(MAKE) -C oracle

# ASan + UBSan over the host half of the library (PLY / JSON / PNG parsers behind the C ABI): no GPU, no hipcc.
# tests/sanitize/host_sanitize_driver.cpp stubs the device entry points and replays the golden files + hostile inputs.
SAN = build/host_sanitize
HOSTSRC = splatapult_amd/host/gaussian_scene.cpp splatapult_amd/host/scene_config.cpp splatapult_amd/host/point_scene.cpp
sanitize: $(HOSTSRC) tests/sanitize/host_sanitize_driver.cpp include/msplat.h include/msplat_debug.h
	mkdir -p build build/sanitize_scratch
	$(CXX) -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -I. \
	    $(HOSTSRC) tests/sanitize/host_sanitize_driver.cpp -o $(SAN)
	ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1 $(SAN) tests/golden build/sanitize_scratch
	rm -rf build/sanitize_scratch

clean:
	rm -rf build $(LIB)

.PHONY: all examples oracle clean sanitize
