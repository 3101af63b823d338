# Build of the MI355X-native predicate evaluator.  Everything is built in-tree (the .so files
# travel to the GPU box with the gpurun snapshot; they are git-ignored).
#
#   make            -> lib (HIP C-ABI library) + host (C++ host mirror) + oracle (test-only CPU restatement)
#   make lib        -> kube_scheduler_rs_reference_amd/libksched_hip.so     hipcc, gfx950 only
#   make host       -> kube_scheduler_rs_reference_amd/libksched_host.so    g++, links libksched_hip.so; + tests/cpp/host_tests
#   make oracle     -> oracle/liboracle.so                                  gcc, test infrastructure only
#   make tools      -> tools/pmc_calib                                      hipcc: calibration kernels for the HBM counters
#   make sanitize   -> build/san/host_tests_{asan,tsan}                     the host mirror + its C++ tests under AddressSanitizer + UBSan, and under ThreadSanitizer
#                      (CPU sanitizers over HOST code only; `tools/sanitize.sh` runs them: the CPU half here, the device half on a GPU box)
#   make test-lib   -> tests/cpp/hooks/libksched_hip.so                     the SAME object code + tests/cpp/test_hooks.cpp: the only build in which
#                      $KSCHED_TEST_HOOKS=1 switches on the RCCL stand-in, fault injection and the k-replica shard (the shipped library has none of it)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
CC      ?= gcc
ARCH    ?= gfx950
PKG     := kube_scheduler_rs_reference_amd
CSRC    := $(PKG)/csrc
HOST    := $(PKG)/host

HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed
CXXFLAGS := -O2 -std=c++17 -fPIC -Wall -Wextra -Iinclude
CFLAGS   := -O2 -std=c11 -fPIC -Wall -Wextra -fopenmp

LIB_HIP  := $(PKG)/libksched_hip.so
LIB_HOST := $(PKG)/libksched_host.so
LIB_ORA  := oracle/liboracle.so

HOST_SRCS := $(wildcard $(HOST)/*.cpp)
HOST_HDRS := $(wildcard $(HOST)/*.hpp) include/ksched.h
HOST_TEST := tests/cpp/host_tests
OBJ_TOOL := tests/cpp/objects_eval
FAKE_RCCL := tests/cpp/libfake_rccl.so
INDEX_TEST := tests/cpp/index_tests

PMC_CALIB := tools/pmc_calib

LIB_OBJ  := $(CSRC)/ksched_api.o
LIB_HIP_TEST := tests/cpp/hooks/libksched_hip.so

.PHONY: all lib host oracle tools clean test-lib sanitize
all: lib test-lib host oracle tools

# kernels of KNOWN byte counts for calibrating the HBM counters (bench.py --live-traffic, tools/gpu_round.sh pmc)
tools: $(PMC_CALIB)
$(PMC_CALIB): tools/pmc_calib.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ tools/pmc_calib.hip

lib: $(LIB_HIP)
# the kernels are compiled ONCE; the shipped library and the test build are two links of the same object
$(LIB_OBJ): $(CSRC)/ksched_api.hip $(wildcard $(CSRC)/*.hpp) include/ksched.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(CSRC)/ksched_api.hip
$(LIB_HIP): $(LIB_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(LIB_OBJ)
test-lib: $(LIB_HIP_TEST)
$(LIB_HIP_TEST): $(LIB_OBJ) tests/cpp/test_hooks.cpp
	mkdir -p tests/cpp/hooks
	$(CXX) -O2 -std=c++17 -fPIC -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c -o tests/cpp/hooks/test_hooks.o tests/cpp/test_hooks.cpp
	$(HIPCC) --offload-arch=$(ARCH) -shared -Wl,-soname,libksched_hip.so -o $@ $(LIB_OBJ) tests/cpp/hooks/test_hooks.o

host: $(LIB_HOST) $(HOST_TEST) $(INDEX_TEST) $(OBJ_TOOL) $(FAKE_RCCL) $(LIB_HIP_TEST)
# TEST-ONLY stand-in for librccl (n ranks on one GPU; loaded only with KSCHED_TEST_HOOKS=1 + KSCHED_RCCL_LIB, see csrc/comm_rccl.hpp)
$(FAKE_RCCL): tests/cpp/fake_rccl.cpp
	$(CXX) -O2 -std=c++17 -fPIC -Wall -Wextra -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -shared -o $@ tests/cpp/fake_rccl.cpp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -lrt -lpthread
# host-only check of the bitmap index arithmetic (no GPU, no HIP runtime call): tests/test_index_host.py runs it
$(INDEX_TEST): tests/cpp/index_tests.cpp $(CSRC)/tile_index.hpp
	$(CXX) -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -o $@ tests/cpp/index_tests.cpp
$(LIB_HOST): $(HOST_SRCS) $(HOST_HDRS) $(LIB_HIP)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRCS) -L$(PKG) -lksched_hip -Wl,-rpath,'$$ORIGIN' -lpthread
# C++ tests of the host mirror (tests/cpp/host_tests.cpp; driven by tests/test_host_mirror.py)
$(HOST_TEST): tests/cpp/host_tests.cpp $(LIB_HOST) $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o $@ tests/cpp/host_tests.cpp -L$(PKG) -lksched_host -lksched_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,/opt/rocm/lib -lpthread -ldl

# objects JSON -> host encoder -> device, printed for the Python parity tests (tests/test_gpu_objects.py)
$(OBJ_TOOL): tests/cpp/objects_eval.cpp tests/cpp/json_min.hpp $(LIB_HOST) $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -o $@ tests/cpp/objects_eval.cpp -L$(PKG) -lksched_host -lksched_hip -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -lpthread

# Host-side sanitizer builds (g++; the evaluator library they link is the ordinary test build: device code is not instrumented)
SAN_DIR  := build/san
SAN_SRCS := tests/cpp/host_tests.cpp $(HOST_SRCS)
SAN_FLAGS := -O1 -g -std=c++17 -fPIC -Wall -Wextra -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -fno-omit-frame-pointer
SAN_LINK := -Ltests/cpp/hooks -lksched_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../../tests/cpp/hooks' -Wl,-rpath,/opt/rocm/lib -lpthread -ldl
SAN_OBJ_SRCS := tests/cpp/objects_eval.cpp $(HOST_SRCS)
sanitize: $(SAN_DIR)/host_tests_asan $(SAN_DIR)/host_tests_tsan $(SAN_DIR)/objects_eval_asan $(SAN_DIR)/objects_eval_tsan
$(SAN_DIR)/asan/%.o: %.cpp $(HOST_HDRS)
	mkdir -p $(dir $@)
	$(CXX) $(SAN_FLAGS) -fsanitize=address,undefined -c -o $@ $<
$(SAN_DIR)/tsan/%.o: %.cpp $(HOST_HDRS)
	mkdir -p $(dir $@)
	$(CXX) $(SAN_FLAGS) -fsanitize=thread -c -o $@ $<
$(SAN_DIR)/asan/tests/cpp/objects_eval.o $(SAN_DIR)/tsan/tests/cpp/objects_eval.o: tests/cpp/json_min.hpp
# the objects -> reconcile_batch -> POST sink -> snapshot loop at any size (tools/host_loop.py with OBJECTS_EVAL_BIN): the threaded path at production batch sizes
$(SAN_DIR)/objects_eval_asan: $(SAN_OBJ_SRCS:%.cpp=$(SAN_DIR)/asan/%.o) $(LIB_HIP_TEST)
	$(CXX) -fsanitize=address,undefined -o $@ $(SAN_OBJ_SRCS:%.cpp=$(SAN_DIR)/asan/%.o) $(SAN_LINK)
$(SAN_DIR)/objects_eval_tsan: $(SAN_OBJ_SRCS:%.cpp=$(SAN_DIR)/tsan/%.o) $(LIB_HIP_TEST)
	$(CXX) -fsanitize=thread -o $@ $(SAN_OBJ_SRCS:%.cpp=$(SAN_DIR)/tsan/%.o) $(SAN_LINK)
$(SAN_DIR)/host_tests_asan: $(SAN_SRCS:%.cpp=$(SAN_DIR)/asan/%.o) $(LIB_HIP_TEST)
	$(CXX) -fsanitize=address,undefined -o $@ $(SAN_SRCS:%.cpp=$(SAN_DIR)/asan/%.o) $(SAN_LINK)
$(SAN_DIR)/host_tests_tsan: $(SAN_SRCS:%.cpp=$(SAN_DIR)/tsan/%.o) $(LIB_HIP_TEST)
	$(CXX) -fsanitize=thread -o $@ $(SAN_SRCS:%.cpp=$(SAN_DIR)/tsan/%.o) $(SAN_LINK)

oracle: $(LIB_ORA)
$(LIB_ORA): oracle/oracle.c oracle/oracle.h
	$(CC) $(CFLAGS) -shared -o $@ oracle/oracle.c

clean:
	rm -f $(LIB_OBJ) $(LIB_HIP_TEST) tests/cpp/hooks/test_hooks.o $(LIB_HIP) $(LIB_HOST) $(LIB_ORA) $(HOST_TEST) $(INDEX_TEST) $(OBJ_TOOL) $(FAKE_RCCL) $(PMC_CALIB)
