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
    {"wgrad_xcd", "MIA_WGRAD_XCD", 1, 0, 1, {1}},
    {"wgrad_dma", "MIA_WGRAD_DMA", 1, 0, 1, {1}},
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
    {"f32_split", "MIA_F32_SPLIT", 2, 0, 2, {2}},
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

int idx(const char* name) {
  for (int i = 0; i < N_OPT; ++i)
    if (strcmp(name, g_table[i].name) == 0) return i;
  return -1;
}

}  // namespace

MiaOptions mia_options() {
  std::call_once(g_env_once, read_env);
  // table positions resolved once (the table is small and fixed)
  static const int i_conv_xcd = idx("conv_xcd"), i_conv64 = idx("conv64"), i_conv_bt = idx("conv_bt"), i_wgrad_xcd = idx("wgrad_xcd"),
                   i_wgrad_dma = idx("wgrad_dma"), i_stream_blocks = idx("stream_blocks"), i_stem_mfma = idx("stem_mfma"),
                   i_conv_bt_order = idx("conv_bt_order"), i_wgrad_bt = idx("wgrad_bt"), i_conv64_dma = idx("conv64_dma"),
                   i_conv_s2_wide = idx("conv_s2_wide"), i_conv_pw = idx("conv_pw"), i_conv_pw_s2 = idx("conv_pw_s2"), i_wgrad_t2 = idx("wgrad_t2"),
                   i_reserve_cus = idx("reserve_cus"), i_f32_split = idx("f32_split");
  MiaOptions o;
  o.conv_xcd = get(i_conv_xcd); o.conv64 = get(i_conv64); o.conv_bt = get(i_conv_bt); o.wgrad_xcd = get(i_wgrad_xcd);
  o.wgrad_dma = get(i_wgrad_dma); o.stream_blocks = get(i_stream_blocks); o.stem_mfma = get(i_stem_mfma); o.conv_bt_order = get(i_conv_bt_order);
  o.wgrad_bt = get(i_wgrad_bt); o.conv64_dma = get(i_conv64_dma); o.conv_s2_wide = get(i_conv_s2_wide); o.conv_pw = get(i_conv_pw);
  o.conv_pw_s2 = get(i_conv_pw_s2); o.wgrad_t2 = get(i_wgrad_t2); o.reserve_cus = get(i_reserve_cus); o.f32_split = get(i_f32_split);
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
