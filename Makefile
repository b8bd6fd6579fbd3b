# Build everything in-tree.  `make` = product libraries + oracle; `make ref` = reference build (only where
# /root/reference exists).
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
CXX       ?= g++
ARCH      ?= gfx950
LIBDIR    := raytracing_amd/lib
CSRC      := raytracing_amd/csrc
HOST      := raytracing_amd/host

# -ffp-contract=off: hit/miss decisions must be bit-identical with the CPU renderer (no FMA contraction).
HIPFLAGS  := -std=c++17 -O3 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wno-unused-result
CXXFLAGS  := -std=c++17 -O2 -fPIC -ffp-contract=off -Iinclude

all: lib host oracle

lib: $(LIBDIR)/librt_amd.so
host: $(LIBDIR)/librt_host.so
oracle:
	$(MAKE) -C oracle oracle
ref:
	$(MAKE) -C oracle ref

# every header of csrc/ is a prerequisite: the traversal code lives in headers that rt_kernels.hip includes
$(LIBDIR)/librt_amd.so: $(CSRC)/rt_kernels.hip $(wildcard $(CSRC)/*.h) $(CSRC)/rt_host_side.cpp include/rt_amd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/rt_kernels.hip -x hip $(CSRC)/rt_host_side.cpp

$(LIBDIR)/librt_host.so: $(wildcard $(HOST)/*.cpp) $(wildcard $(HOST)/*.h) include/rt_amd.h $(LIBDIR)/librt_amd.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(wildcard $(HOST)/*.cpp) -L$(LIBDIR) -lrt_amd -lz -Wl,-rpath,'$$ORIGIN'

clean:
	rm -f $(LIBDIR)/*.so
	$(MAKE) -C oracle clean

.PHONY: all lib host oracle ref clean
