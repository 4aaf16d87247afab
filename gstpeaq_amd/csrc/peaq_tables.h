// peaq_tables.h -- host-side construction of the constant tables.
#pragma once
#include "peaq_device.h"

namespace peaq {
void build_common_tables(CommonTables& c);
void build_fft_band_tables(int bands, BandTables& t);           // 109 (basic) or 55 (advanced)
void build_fb_band_tables(BandTables& t, FbTables& fb);         // 40-band filter bank
double fb_tables_selfcheck();                                    // host-only check of the FP64 engine's tables
double fft_level_factor(double playback_level_db);
double fb_level_factor(double playback_level_db);
}  // namespace peaq
