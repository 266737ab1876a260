// Host-emulator flavour of libstcat_hip: same kernel sources, same C ABI.
#include "hip_emu.h"
EMU_DEFINE_GLOBALS
#include "../../stcat_amd/csrc/stcat_capi.hip"
