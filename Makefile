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
OBJS := $(HIP_SRCS:.hip=.o) $(CPP_SRCS:.cpp=.o)
LIB := speck_amd/libspeck_amd.so

all: $(LIB) oracle apps

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/speck_c_api.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(CSRC)/%.o: $(CSRC)/%.cpp $(wildcard $(CSRC)/*.hpp) include/speck_c_api.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -s -C oracle

apps: $(LIB)
	@if [ -f apps/runspECK.cpp ]; then \
	  $(HIPCC) $(HIPFLAGS) -x hip apps/runspECK.cpp -o apps/runspECK -Lspeck_amd -lspeck_amd \
	      -L/opt/rocm/lib -lrocsparse -Wl,-rpath,'$$ORIGIN/../speck_amd' -Wl,-rpath,/opt/rocm/lib; fi

clean:
	rm -f $(OBJS) $(LIB) apps/runspECK
	$(MAKE) -s -C oracle clean
.PHONY: all oracle apps clean
