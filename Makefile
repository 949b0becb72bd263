# Build of the MI355X-native SpGEMM backend: hipcc cross-compiles gfx950 without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := speck_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
HIPFLAGS += $(EXTRA)
ifdef PHASE_CLOCKS
HIPFLAGS += -DSPECK_PHASE_CLOCKS
endif
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
CPP_SRCS := $(wildcard $(CSRC)/*.cpp)
# make ASAN=1: the HOST side of every object under AddressSanitizer + UndefinedBehaviorSanitizer (the device code is
# untouched), into a library of its own -- speck_amd/libspeck_amd_asan.so, loaded with SPECK_LIB=... and the sanitizer
# runtime preloaded (scripts/asan_suite.sh).  SURVEY.md 5: the reference has no sanitizer pass either.
# make UBSAN=1: UndefinedBehaviorSanitizer alone (libspeck_amd_ubsan.so) -- what a process that talks to a GPU can run
# under: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and aborts without an xnack+ device build.
ifdef ASAN
OSUF := .asan.o
LIB := speck_amd/libspeck_amd_asan.so
SANFLAGS := -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -g -shared-libsan
HIPFLAGS += $(SANFLAGS)
LDSAN := -fsanitize=address,undefined -shared-libsan
else ifdef UBSAN
OSUF := .ubsan.o
LIB := speck_amd/libspeck_amd_ubsan.so
# (-fno-sanitize=function: with the function-type check a kernel launched through a function-pointer variable --
#  `auto k = kernel<...>; hipLaunchKernelGGL(k, ...)` -- pushes its launch configuration and never reaches its stub)
SANFLAGS := -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize=function -Xarch_host -fno-omit-frame-pointer -Xarch_host -g -shared-libsan
HIPFLAGS += $(SANFLAGS)
LDSAN := -fsanitize=undefined -shared-libsan
else ifdef POISON
# make POISON=1: every kernel poisons its LDS first (device_common.hpp, poison_lds) -- libspeck_amd_poison.so
OSUF := .poison.o
LIB := speck_amd/libspeck_amd_poison.so
HIPFLAGS += -DSPECK_POISON_LDS
LDSAN :=
else
OSUF := .o
LIB := speck_amd/libspeck_amd.so
LDSAN :=
endif
OBJS := $(HIP_SRCS:.hip=$(OSUF)) $(CPP_SRCS:.cpp=$(OSUF))

# (the variant libraries -- ASAN / UBSAN / POISON -- are built alone: the driver and the oracle belong to the plain build)
ifeq ($(OSUF),.o)
all: $(LIB) oracle apps
else
all: $(LIB)
endif

$(CSRC)/%$(OSUF): $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/speck_c_api.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/%$(OSUF): $(CSRC)/%.cpp $(wildcard $(CSRC)/*.hpp) include/speck_c_api.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(LDSAN) -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle

apps: $(LIB)
	@if [ -f apps/runspECK.cpp ]; then \
	  $(HIPCC) $(HIPFLAGS) -x hip apps/runspECK.cpp -o apps/runspECK -Lspeck_amd -lspeck_amd \
	      -L/opt/rocm/lib -lrocsparse -Wl,-rpath,'$$ORIGIN/../speck_amd' -Wl,-rpath,/opt/rocm/lib; fi

clean:
	rm -f $(OBJS) $(LIB) $(CSRC)/*.asan.o $(CSRC)/*.ubsan.o $(CSRC)/*.poison.o speck_amd/libspeck_amd_asan.so speck_amd/libspeck_amd_ubsan.so speck_amd/libspeck_amd_poison.so apps/runspECK
	$(MAKE) -s -C oracle clean
.PHONY: all oracle apps clean
