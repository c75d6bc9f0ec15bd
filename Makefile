# Build of the product libraries and the test oracle.  `python -c "import __graft_entry__ as g; g.build()"` runs this.
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
ROOT     := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
LIBDIR   ?= actinon_amd/lib
BUILD    ?= build
CFLAGS   := -O2 -fPIC -std=gnu11 -Wall -Wno-unused-function -ffp-contract=off -Iinclude
SHADE_WAVES ?= 4
WALK_WAVES  ?= 4
# k_walk: 4 waves per SIMD (128 VGPRs) since round 4.  Rounds 2 - 3 ran it at 2 (256 VGPRs): with every operand of every pair
# expanded four times in line the kernel was 1.4 MB of code and spilled ~800 registers at 128.  With the operand loops of
# pair_hit, the roughness perturbation as a real call and one copy of the root traversal it is 140 KB, spills ~160 at 128, and
# the two extra waves per SIMD hide more latency than the spills add: 1080p 57.6 -> 53.0 ms, diamond -11 % (profiles/r04/)
TRACE_WAVES ?= 4
EXTRA_DEFS ?=
OPT ?= -O3
HIPFLAGS := $(EXTRA_DEFS) -DACN_SHADE_WAVES=$(SHADE_WAVES) -DACN_WALK_WAVES=$(WALK_WAVES) -DACN_TRACE_WAVES=$(TRACE_WAVES) $(OPT) -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -Iactinon_amd/csrc -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result

all: hip host oracle cli

hip: $(LIBDIR)/libactinon_hip.so
host: $(LIBDIR)/libactinon_host.so
oracle: oracle/libacn_oracle.so oracle/libacn_oracle_libm.so
cli: actinon_amd/bin/actinon_hip

# one object per kernel family: `make -j` compiles them side by side (acn_launch.h)
HIP_UNITS := actinon_hip k_shade_64 k_shade_16 k_shade_4 k_shade_1 k_walk_lds k_walk_glb k_walk_count k_walk_count_prune k_hard_shadow k_hard_path
HIP_OBJS  := $(addprefix $(BUILD)/,$(addsuffix .o,$(HIP_UNITS)))
# the two production units of k_walk are scheduled for instruction-level parallelism: __launch_bounds__ fixes their occupancy
# (2 waves per SIMD), which is what the default strategy schedules for (hanging_lamp 600x800 -3 %, everything else equal:
# profiles/r03/ab_walk_max_ilp_scheduling.txt; same bits: scheduling does not reassociate)
WALK_SCHED ?= -mllvm -amdgpu-sched-strategy=max-ilp
$(BUILD)/k_walk_lds.o $(BUILD)/k_walk_glb.o: HIPFLAGS += $(WALK_SCHED)
$(BUILD)/%.o: actinon_amd/csrc/%.hip $(wildcard actinon_amd/csrc/*.h) include/actinon_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(LIBDIR)/libactinon_hip.so: $(HIP_OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIP_OBJS)

$(LIBDIR)/libactinon_host.so: actinon_amd/host/acn_scene.c actinon_amd/host/acn_driver.c actinon_amd/host/acn_scenes.c actinon_amd/host/acn_interp.c include/acn_scene.h include/acn_interp.h include/actinon_hip.h $(LIBDIR)/libactinon_hip.so
	$(CC) $(CFLAGS) -shared -o $@ actinon_amd/host/acn_scene.c actinon_amd/host/acn_driver.c actinon_amd/host/acn_scenes.c actinon_amd/host/acn_interp.c -L$(LIBDIR) -lactinon_hip -lm -Wl,-rpath,'$$ORIGIN'

actinon_amd/bin/actinon_hip: tools/actinon_hip.c include/acn_interp.h $(LIBDIR)/libactinon_host.so
	@mkdir -p actinon_amd/bin
	$(CC) $(CFLAGS) -o $@ tools/actinon_hip.c -L$(LIBDIR) -lactinon_host -lactinon_hip -lm -Wl,-rpath,'$$ORIGIN/../lib'

# the two prebuilt oracles travel to the GPU box: a portable ISA level (AVX2 + FMA), not the build container's -march=native
# (bench.py compiles a third build on the box itself with the reference's flags for the CPU baseline)
ORACLE_MARCH ?= x86-64-v3
oracle/libacn_oracle.so: oracle/acn_oracle.c oracle/acn_oracle.h actinon_amd/csrc/acn_detmath.h actinon_amd/csrc/acn_costs.h include/actinon_hip.h
	$(CC) $(CFLAGS) -march=$(ORACLE_MARCH) -Ioracle -shared -o $@ oracle/acn_oracle.c -lm -lpthread

oracle/libacn_oracle_libm.so: oracle/acn_oracle.c oracle/acn_oracle.h actinon_amd/csrc/acn_costs.h include/actinon_hip.h
	$(CC) $(CFLAGS) -march=$(ORACLE_MARCH) -DACN_ORACLE_LIBM -Ioracle -shared -o $@ oracle/acn_oracle.c -lm -lpthread

clean:
	rm -rf build; rm -f $(LIBDIR)/*.so oracle/*.so actinon_amd/bin/actinon_hip

.PHONY: all hip host oracle cli clean
