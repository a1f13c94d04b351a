// Option table of libmia_hip (see options.h): environment read once, atomics, mia_set_option / mia_get_option.
#include "options.h"
#include "common.h"
#include <atomic>
#include <mutex>
#include <stdlib.h>

namespace {

struct Entry { const char* name; const char* env; int def; int lo; int hi; std::atomic<int> value; };

Entry g_table[] = {
    {"conv_xcd", "MIA_CONV_XCD", 1, 0, 1, {1}},
    {"conv64", "MIA_CONV64", 1, 0, 1, {1}},
    {"conv_bt", "MIA_CONV_BT", 1, 0, 1, {1}},
    {"conv_mt8", "MIA_CONV_MT8", 0, 0, 1, {0}},
    {"conv64_blocks", "MIA_CONV64_BLOCKS", 0, 0, 1 << 20, {0}},
    {"wgrad_xcd", "MIA_WGRAD_XCD", 1, 0, 1, {1}},
    {"wgrad_dma", "MIA_WGRAD_DMA", 1, 0, 1, {1}},
    {"wgrad_tab", "MIA_WGRAD_TAB", 1, 0, 1, {1}},
    {"wgrad_w8", "MIA_WGRAD_W8", 1, 0, 1, {1}},
    {"stream_blocks", "MIA_STREAM_BLOCKS", 32768, 256, 1 << 24, {32768}},
    {"stem_mfma", "MIA_STEM_MFMA", 1, 0, 1, {1}},
    {"conv_bt_order", "MIA_CONV_BT_ORDER", 1, 0, 1, {1}},
    {"wgrad_bt", "MIA_WGRAD_BT", 1, 0, 1, {1}},
    {"conv64_dma", "MIA_CONV64_DMA", 1, 0, 2, {1}},
    {"conv_s2_wide", "MIA_CONV_S2_WIDE", 1, 0, 2, {1}},
    {"conv_pw", "MIA_CONV_PW", 1, 0, 1, {1}},
    {"conv_pw_s2", "MIA_CONV_PW_S2", 1, 0, 2, {1}},
    {"wgrad_t2", "MIA_WGRAD_T2", 1, 0, 1, {1}},
    {"reserve_cus", "MIA_RESERVE_CUS", 0, 0, 64, {0}},
    {"conv_t3_wide", "MIA_CONV_T3_WIDE", 0, 0, 1, {0}},
    {"wgrad_narrow", "MIA_WGRAD_NARROW", 0, 0, 1, {0}},
    {"f32_split", "MIA_F32_SPLIT", 1, 0, 1, {1}},
    {"conv_pw_t3", "MIA_CONV_PW_T3", 0, 0, 1, {0}},
    {"conv64_wino", "MIA_CONV64_WINO", 0, 0, 1, {0}},
};
constexpr int N_OPT = (int)(sizeof(g_table) / sizeof(g_table[0]));
std::once_flag g_env_once;

int clampv(const Entry& e, int v) { return v < e.lo ? e.lo : (v > e.hi ? e.hi : v); }

void read_env() {
  for (int i = 0; i < N_OPT; ++i) {
    const char* s = getenv(g_table[i].env);
    if (s && *s) g_table[i].value.store(clampv(g_table[i], atoi(s)), std::memory_order_relaxed);
  }
}

Entry* find(const char* name) {
  std::call_once(g_env_once, read_env);
  for (int i = 0; i < N_OPT; ++i)
    if (strcmp(name, g_table[i].name) == 0) return &g_table[i];
  return nullptr;
}

int get(int i) { return g_table[i].value.load(std::memory_order_relaxed); }

}  // namespace

MiaOptions mia_options() {
  std::call_once(g_env_once, read_env);
  MiaOptions o;
  o.conv_xcd = get(0); o.conv64 = get(1); o.conv_bt = get(2); o.conv_mt8 = get(3); o.conv64_blocks = get(4);
  o.wgrad_xcd = get(5); o.wgrad_dma = get(6); o.wgrad_tab = get(7); o.wgrad_w8 = get(8); o.stream_blocks = get(9);
  o.stem_mfma = get(10); o.conv_bt_order = get(11); o.wgrad_bt = get(12); o.conv64_dma = get(13); o.conv_s2_wide = get(14); o.conv_pw = get(15); o.conv_pw_s2 = get(16); o.wgrad_t2 = get(17); o.reserve_cus = get(18); o.conv_t3_wide = get(19); o.wgrad_narrow = get(20); o.f32_split = get(21); o.conv_pw_t3 = get(22); o.conv64_wino = get(23);
  return o;
}

extern "C" int mia_set_option(const char* name, int value) {
  MIA_CHECK_ARG(name != nullptr, "mia_set_option: null name");
  Entry* e = find(name);
  MIA_CHECK_ARG(e != nullptr, "mia_set_option: unknown option '%s'", name);
  e->value.store(clampv(*e, value), std::memory_order_relaxed);
  return MIA_OK;
}

extern "C" int mia_get_option(const char* name, int* value) {
  MIA_CHECK_ARG(name != nullptr && value != nullptr, "mia_get_option: null argument");
  Entry* e = find(name);
  MIA_CHECK_ARG(e != nullptr, "mia_get_option: unknown option '%s'", name);
  *value = e->value.load(std::memory_order_relaxed);
  return MIA_OK;
}
