# Build recipe: the product library (HIP, gfx950 only), the host-side facade demo, the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
PKG := voxel-cone-tracing_amd
CSRC := $(PKG)/csrc
# -ffp-contract=off: the kernels mirror the oracle's fp32 operation order (explicit fmaf only).
# -fno-slp-vectorize: packed fp32 (v_pk_fma_f32 ...) issues at half rate on gfx950, so SLP packing buys
#   nothing and costs v_mov shuffles (trace kernel 1.01 -> 0.87 ms, profiles/r01c).
HIPFLAGS := -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -Wall -Wno-unused-function
OBJS := $(CSRC)/vct_capi.o $(CSRC)/vct_trace.o $(CSRC)/vct_volume.o $(CSRC)/vct_voxelize.o $(CSRC)/vct_raster.o $(CSRC)/vct_multi.o
LIB := $(PKG)/libvct_amd.so

HOSTLIB := $(PKG)/libvct_host.so

all: lib host oracle demo

host: $(HOSTLIB)

$(HOSTLIB): $(PKG)/host/vct_host.cpp $(PKG)/host/vct_host.h $(PKG)/host/vct_image.h
	g++ -O2 -std=c++17 -ffp-contract=off -fPIC -Wall -Wextra -shared -o $@ $(PKG)/host/vct_host.cpp

lib: $(LIB)

# headless caller written against the facade header (the reference application's call sequence)
DEMO := $(PKG)/vct_demo
demo: $(DEMO)
$(DEMO): $(PKG)/host/demo_main.cpp $(PKG)/host/Voxel_Cone_Tracing.h include/vct.h $(LIB) $(HOSTLIB)
	g++ -O2 -std=c++17 -Wall -Wextra -o $@ $(PKG)/host/demo_main.cpp -L$(PKG) -lvct_amd -lvct_host \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/vct_internal.h $(CSRC)/vct_layout.h $(CSRC)/vct_ctx.h $(CSRC)/vct_divisors.h include/vct.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB) $(HOSTLIB) $(DEMO)
	$(MAKE) -C oracle clean

.PHONY: all lib host oracle demo clean
