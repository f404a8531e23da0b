/*
 * ouster_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference algorithms for
 *   packet-format field decode -> LidarFrame, destagger, make_xyz_lut, cartesian.
 * See ouster_oracle.h for the usage rule (tests / smoke / cpu_baseline only).
 * All citations are relative to /root/reference.
 */
#include "ouster_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* type helpers: ouster_core/src/chanfield.cpp:54-89                         */
/* ------------------------------------------------------------------------- */
size_t ora_type_size(int t) {
    switch (t) {
        case ORA_I8: case ORA_U8: return 1;
        case ORA_I16: case ORA_U16: case ORA_F16: return 2;
        case ORA_I32: case ORA_U32: case ORA_F32: return 4;
        case ORA_I64: case ORA_U64: case ORA_F64: return 8;
        default: return 0;
    }
}

static uint64_t type_mask(int t) {
    switch (ora_type_size(t)) {
        case 1: return 0xffull;
        case 2: return 0xffffull;
        case 4: return 0xffffffffull;
        case 8: return ~0ull;
        default: return 0;
    }
}

/* ------------------------------------------------------------------------- */
/* field_info(): ouster_core/src/parsing.cpp:57-122                          */
/* ------------------------------------------------------------------------- */
int ora_field_info(uint64_t bit_start, uint64_t bit_size, uint64_t upshift,
                   uint64_t max_length, uint64_t num_elements, ora_fdi* out) {
    ora_fdi f;
    memset(&f, 0, sizeof f);
    uint64_t needs_bits = bit_size + upshift;
    if (needs_bits > 64) return -1; /* "value cannot store more than 64 bits" */

    f.offset = bit_start / 8;
    bit_start %= 8;
    for (uint64_t i = bit_start; i < bit_start + bit_size; ++i)
        f.mask |= 1ull << i;
    f.shift = (int)bit_start - (int)upshift;
    f.num_elements = (int)num_elements;

    uint64_t size_bytes = needs_bits / 8 + ((needs_bits % 8) ? 1 : 0);
    size_bytes /= num_elements;
    switch (size_bytes) {
        case 1: f.ty_tag = ORA_U8; break;
        case 2: f.ty_tag = ORA_U16; break;
        case 3: case 4: f.ty_tag = ORA_U32; break;
        case 5: case 6: case 7: case 8: f.ty_tag = ORA_U64; break;
        default: f.ty_tag = ORA_VOID;
    }
    if (max_length > 0) {
        if (f.offset + size_bytes > max_length) return -2; /* read past end */
        int need = (int)f.offset + 8 - (int)max_length;
        if (need > 0) {
            f.offset -= (uint64_t)need;
            f.mask <<= need * 8;
            f.shift += need * 8;
        }
    }
    *out = f;
    return 0;
}

/* FieldDecodeInfo::get / set: include/ouster/core/field_decode_info.h:41-78.
 * The caller truncates the returned word to its destination width
 * (the reference memcpy's the low sizeof(T) bytes, little endian). */
uint64_t ora_fdi_get(const ora_fdi* f, const uint8_t* buf) {
    uint64_t word;
    memcpy(&word, buf + f->offset, 8);
    word &= f->mask;
    if (f->shift > 0) word >>= f->shift;
    else if (f->shift < 0) word <<= -f->shift;
    return word;
}

void ora_fdi_set(const ora_fdi* f, uint8_t* buf, uint64_t word) {
    if (f->shift > 0) word <<= f->shift;
    if (f->shift < 0) word >>= -f->shift;
    word &= f->mask;
    uint64_t cur;
    memcpy(&cur, buf + f->offset, 8);
    cur &= ~f->mask;
    cur |= word;
    memcpy(buf + f->offset, &cur, 8);
}

/* impl::get_value_mask: ouster_core/src/parsing.cpp:139-157 */
uint64_t ora_value_mask(const ora_fdi* f) {
    uint64_t tm = type_mask(f->ty_tag);
    uint64_t m = f->mask;
    if (m == 0) m = tm;
    if (f->shift > 0) m >>= f->shift;
    if (f->shift < 0) m <<= -f->shift;
    return m & tm;
}

/* ------------------------------------------------------------------------- */
/* profile tables: ouster_core/src/parsing.cpp:170-363                       */
/* ------------------------------------------------------------------------- */
typedef struct {
    const char* name;
    uint16_t bit_start, bit_size, upshift, num_elements;
} field_spec;

#define RAW1 {"RAW32_WORD1", 0, 32, 0, 1}
#define RAW2 {"RAW32_WORD2", 32, 32, 0, 1}
#define RAW3 {"RAW32_WORD3", 64, 32, 0, 1}
#define RAW4 {"RAW32_WORD4", 96, 32, 0, 1}
#define RAW5 {"RAW32_WORD5", 128, 32, 0, 1}

static const field_spec T_LEGACY[] = { /* parsing.cpp:170-179 */
    {"RANGE", 0, 20, 0, 1}, {"FLAGS", 28, 4, 0, 1}, {"REFLECTIVITY", 32, 8, 0, 1},
    {"SIGNAL", 48, 16, 0, 1}, {"NEAR_IR", 64, 16, 0, 1}, RAW1, RAW2, RAW3};
static const field_spec T_LB[] = { /* :181-187 */
    {"RANGE", 0, 15, 3, 1}, {"FLAGS", 15, 1, 0, 1}, {"REFLECTIVITY", 16, 8, 0, 1},
    {"NEAR_IR", 24, 8, 4, 1}, RAW1};
static const field_spec T_LB_WIN[] = { /* :189-195 */
    {"RANGE", 0, 15, 3, 1}, {"FLAGS", 15, 1, 0, 1}, {"REFLECTIVITY", 16, 8, 0, 1},
    {"WINDOW", 24, 8, 0, 1}, RAW1};
static const field_spec T_RGB[] = { /* :197-211 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 24, 8, 0, 1},
    {"SIGNAL", 32, 16, 0, 1}, {"NEAR_IR", 48, 16, 0, 1}, {"R", 64, 16, 0, 1},
    {"G", 80, 16, 0, 1}, {"B", 96, 16, 0, 1}, {"RGB", 64, 48, 0, 3},
    RAW1, RAW2, RAW3, RAW4};
static const field_spec T_DUAL_RGB[] = { /* :213-232 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 24, 8, 0, 1},
    {"RANGE2", 32, 19, 0, 1}, {"FLAGS2", 51, 5, 0, 1}, {"REFLECTIVITY2", 56, 8, 0, 1},
    {"SIGNAL", 64, 16, 0, 1}, {"SIGNAL2", 80, 16, 0, 1}, {"NEAR_IR", 96, 16, 0, 1},
    {"R", 112, 16, 0, 1}, {"G", 128, 16, 0, 1}, {"B", 144, 16, 0, 1},
    {"RGB", 112, 48, 0, 3}, RAW1, RAW2, RAW3, RAW4, RAW5};
static const field_spec T_DUAL[] = { /* :234-249 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 24, 8, 0, 1},
    {"RANGE2", 32, 19, 0, 1}, {"FLAGS2", 51, 5, 0, 1}, {"REFLECTIVITY2", 56, 8, 0, 1},
    {"SIGNAL", 64, 16, 0, 1}, {"SIGNAL2", 80, 16, 0, 1}, {"NEAR_IR", 96, 16, 0, 1},
    {"WINDOW", 120, 8, 0, 1}, RAW1, RAW2, RAW3, RAW4};
static const field_spec T_SINGLE[] = { /* :251-261 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 32, 8, 0, 1},
    {"SIGNAL", 48, 16, 0, 1}, {"NEAR_IR", 64, 16, 0, 1}, {"WINDOW", 88, 8, 0, 1},
    RAW1, RAW2, RAW3};
static const field_spec T_FIVE[] = { /* :263-278 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 24, 8, 0, 1},
    {"RANGE2", 32, 19, 0, 1}, {"FLAGS2", 51, 5, 0, 1}, {"REFLECTIVITY2", 56, 8, 0, 1},
    {"SIGNAL", 64, 16, 0, 1}, {"SIGNAL2", 80, 16, 0, 1}, {"NEAR_IR", 96, 16, 0, 1},
    RAW1, RAW2, RAW3, RAW4, RAW5};
static const field_spec T_ZM_LB[] = { /* :280-289 */
    {"RANGE", 0, 15, 3, 1}, {"FLAGS", 15, 1, 0, 1}, {"REFLECTIVITY", 16, 8, 0, 1},
    {"NEAR_IR", 24, 8, 4, 1}, {"ZONE_MASK", 32, 16, 0, 1}, {"WINDOW", 48, 8, 0, 1},
    RAW1, RAW2};
static const field_spec T_ZM_SINGLE[] = { /* :291-302 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 32, 8, 0, 1},
    {"WINDOW", 40, 8, 0, 1}, {"SIGNAL", 48, 16, 0, 1}, {"NEAR_IR", 64, 16, 0, 1},
    {"ZONE_MASK", 80, 16, 0, 1}, RAW1, RAW2, RAW3};
static const field_spec T_DUAL_LB[] = { /* :304-315 */
    {"RANGE", 0, 15, 3, 1}, {"FLAGS", 15, 1, 0, 1}, {"REFLECTIVITY", 16, 8, 0, 1},
    {"NEAR_IR", 24, 8, 4, 1}, {"RANGE2", 32, 15, 3, 1}, {"FLAGS2", 47, 1, 0, 1},
    {"REFLECTIVITY2", 48, 8, 0, 1}, {"WINDOW", 56, 8, 0, 1}, RAW1, RAW2};
static const field_spec T_DUAL_ZONE[] = { /* :317-332 */
    {"RANGE", 0, 19, 0, 1}, {"FLAGS", 19, 5, 0, 1}, {"REFLECTIVITY", 24, 8, 0, 1},
    {"RANGE2", 32, 19, 0, 1}, {"FLAGS2", 51, 5, 0, 1}, {"REFLECTIVITY2", 56, 8, 0, 1},
    {"SIGNAL", 64, 16, 0, 1}, {"SIGNAL2", 80, 16, 0, 1}, {"ZONE_MASK", 96, 16, 0, 1},
    {"WINDOW", 120, 8, 0, 1}, RAW1, RAW2, RAW3, RAW4};

#define NSPEC(a) ((int)(sizeof(a) / sizeof((a)[0])))

typedef struct {
    int profile;
    const field_spec* specs;
    int n;
    uint64_t chan_data_size;
} profile_entry;

/* parsing.cpp:337-363 */
static const profile_entry PROFILES[] = {
    {ORA_PROFILE_LEGACY, T_LEGACY, NSPEC(T_LEGACY), 12},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_DUAL, T_DUAL, NSPEC(T_DUAL), 16},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16, T_SINGLE, NSPEC(T_SINGLE), 12},
    {ORA_PROFILE_RNG15_RFL8_NIR8, T_LB, NSPEC(T_LB), 4},
    {ORA_PROFILE_FIVE_WORD_PIXEL, T_FIVE, NSPEC(T_FIVE), 20},
    {ORA_PROFILE_FUSA_RNG15_RFL8_NIR8_DUAL, T_DUAL_LB, NSPEC(T_DUAL_LB), 8},
    {ORA_PROFILE_RNG15_RFL8_NIR8_DUAL, T_DUAL_LB, NSPEC(T_DUAL_LB), 8},
    {ORA_PROFILE_RNG15_RFL8_NIR8_ZONE16, T_ZM_LB, NSPEC(T_ZM_LB), 8},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_ZONE16, T_ZM_SINGLE, NSPEC(T_ZM_SINGLE), 12},
    {ORA_PROFILE_RNG15_RFL8_WIN8, T_LB_WIN, NSPEC(T_LB_WIN), 4},
    {ORA_PROFILE_RNG19_RFL8_SIG16_ZONE16_DUAL, T_DUAL_ZONE, NSPEC(T_DUAL_ZONE), 16},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16, T_RGB, NSPEC(T_RGB), 16},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16_DUAL, T_DUAL_RGB, NSPEC(T_DUAL_RGB), 20},
};

/* default LidarFrame planes per profile: ouster_core/src/lidar_frame.cpp:73-227 */
typedef struct { const char* name; int ty; } slot_spec;
static const slot_spec S_LEGACY[] = {{"RANGE", ORA_U32}, {"SIGNAL", ORA_U16},
    {"NEAR_IR", ORA_U16}, {"REFLECTIVITY", ORA_U8}, {"FLAGS", ORA_U8}};
static const slot_spec S_DUAL[] = {{"RANGE", ORA_U32}, {"RANGE2", ORA_U32},
    {"SIGNAL", ORA_U16}, {"SIGNAL2", ORA_U16}, {"REFLECTIVITY", ORA_U8},
    {"REFLECTIVITY2", ORA_U8}, {"FLAGS", ORA_U8}, {"FLAGS2", ORA_U8},
    {"NEAR_IR", ORA_U16}, {"WINDOW", ORA_U8}};
static const slot_spec S_SINGLE[] = {{"RANGE", ORA_U32}, {"SIGNAL", ORA_U16},
    {"REFLECTIVITY", ORA_U8}, {"FLAGS", ORA_U8}, {"NEAR_IR", ORA_U16},
    {"WINDOW", ORA_U8}};
static const slot_spec S_RGB[] = {{"RANGE", ORA_U32}, {"SIGNAL", ORA_U16},
    {"REFLECTIVITY", ORA_U8}, {"NEAR_IR", ORA_U16}, {"RGB", ORA_F16},
    {"FLAGS", ORA_U8}};
static const slot_spec S_DUAL_RGB[] = {{"RANGE", ORA_U32}, {"RANGE2", ORA_U32},
    {"SIGNAL", ORA_U16}, {"SIGNAL2", ORA_U16}, {"REFLECTIVITY", ORA_U8},
    {"REFLECTIVITY2", ORA_U8}, {"NEAR_IR", ORA_U16}, {"RGB", ORA_F16},
    {"FLAGS", ORA_U8}, {"FLAGS2", ORA_U8}};
static const slot_spec S_LB[] = {{"RANGE", ORA_U32}, {"REFLECTIVITY", ORA_U8},
    {"NEAR_IR", ORA_U16}, {"FLAGS", ORA_U8}};
static const slot_spec S_LB_WIN[] = {{"RANGE", ORA_U32}, {"REFLECTIVITY", ORA_U8},
    {"WINDOW", ORA_U8}, {"FLAGS", ORA_U8}};
static const slot_spec S_ZM_LB[] = {{"RANGE", ORA_U32}, {"REFLECTIVITY", ORA_U8},
    {"NEAR_IR", ORA_U16}, {"FLAGS", ORA_U8}, {"ZONE_MASK", ORA_U16},
    {"WINDOW", ORA_U8}};
static const slot_spec S_ZM_SINGLE[] = {{"RANGE", ORA_U32}, {"SIGNAL", ORA_U16},
    {"REFLECTIVITY", ORA_U8}, {"FLAGS", ORA_U8}, {"NEAR_IR", ORA_U16},
    {"ZONE_MASK", ORA_U16}, {"WINDOW", ORA_U8}};
static const slot_spec S_FIVE[] = {{"RAW32_WORD1", ORA_U32}, {"RAW32_WORD2", ORA_U32},
    {"RAW32_WORD3", ORA_U32}, {"RAW32_WORD4", ORA_U32}, {"RAW32_WORD5", ORA_U32}};
static const slot_spec S_DUAL_LB[] = {{"RANGE", ORA_U32}, {"REFLECTIVITY", ORA_U8},
    {"NEAR_IR", ORA_U16}, {"RANGE2", ORA_U32}, {"REFLECTIVITY2", ORA_U8},
    {"FLAGS", ORA_U8}, {"FLAGS2", ORA_U8}, {"WINDOW", ORA_U8}};
static const slot_spec S_ZM_DUAL[] = {{"RANGE", ORA_U32}, {"RANGE2", ORA_U32},
    {"SIGNAL", ORA_U16}, {"SIGNAL2", ORA_U16}, {"REFLECTIVITY", ORA_U8},
    {"REFLECTIVITY2", ORA_U8}, {"FLAGS", ORA_U8}, {"FLAGS2", ORA_U8},
    {"ZONE_MASK", ORA_U16}, {"WINDOW", ORA_U8}};

typedef struct { int profile; const slot_spec* s; int n; } slots_entry;
#define NS(a) ((int)(sizeof(a) / sizeof((a)[0])))
static const slots_entry SLOTS[] = { /* lidar_frame.cpp:197-226 */
    {ORA_PROFILE_LEGACY, S_LEGACY, NS(S_LEGACY)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_DUAL, S_DUAL, NS(S_DUAL)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16, S_SINGLE, NS(S_SINGLE)},
    {ORA_PROFILE_RNG15_RFL8_NIR8, S_LB, NS(S_LB)},
    {ORA_PROFILE_RNG15_RFL8_WIN8, S_LB_WIN, NS(S_LB_WIN)},
    {ORA_PROFILE_FIVE_WORD_PIXEL, S_FIVE, NS(S_FIVE)},
    {ORA_PROFILE_FUSA_RNG15_RFL8_NIR8_DUAL, S_DUAL_LB, NS(S_DUAL_LB)},
    {ORA_PROFILE_RNG15_RFL8_NIR8_DUAL, S_DUAL_LB, NS(S_DUAL_LB)},
    {ORA_PROFILE_RNG15_RFL8_NIR8_ZONE16, S_ZM_LB, NS(S_ZM_LB)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_ZONE16, S_ZM_SINGLE, NS(S_ZM_SINGLE)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_ZONE16_DUAL, S_ZM_DUAL, NS(S_ZM_DUAL)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16, S_RGB, NS(S_RGB)},
    {ORA_PROFILE_RNG19_RFL8_SIG16_NIR16_RGB16_DUAL, S_DUAL_RGB, NS(S_DUAL_RGB)},
};

/* custom profiles: ouster_core/src/profile_extension.cpp:130-183 */
#define MAX_CUSTOM 16
typedef struct {
    int n_fields;
    ora_field fields[ORA_MAX_FIELDS];
    int32_t slot_types[ORA_MAX_FIELDS];
    uint64_t chan_data_size;
} custom_profile;
static custom_profile g_custom[MAX_CUSTOM];
static int g_n_custom = 0;

int ora_add_custom_profile(const ora_field* fields, int n_fields,
                           uint64_t chan_data_size, const int32_t* slot_types) {
    if (g_n_custom >= MAX_CUSTOM || n_fields > ORA_MAX_FIELDS) return -1;
    custom_profile* c = &g_custom[g_n_custom];
    c->n_fields = n_fields;
    c->chan_data_size = chan_data_size;
    for (int i = 0; i < n_fields; ++i) {
        c->fields[i] = fields[i];
        if (c->fields[i].info.mask == 0) /* profile_extension.cpp:147-149 */
            c->fields[i].info.mask = type_mask(c->fields[i].info.ty_tag);
        if (c->fields[i].info.num_elements == 0) c->fields[i].info.num_elements = 1;
        c->slot_types[i] = slot_types ? slot_types[i] : fields[i].info.ty_tag;
    }
    return ORA_PROFILE_CUSTOM_BASE + g_n_custom++;
}

static int cmp_field_name(const void* a, const void* b) {
    return strcmp(((const ora_field*)a)->name, ((const ora_field*)b)->name);
}

/* PacketFormat::Impl(const DataFormat&): parsing.cpp:453-538 (lidar part),
 * max_frame_id: data_format.cpp:156-161 */
int ora_pf_init(ora_pf* pf, int profile, int header_type, uint32_t h,
                uint32_t cpp, uint32_t w) {
    memset(pf, 0, sizeof *pf);
    int legacy = (profile == ORA_PROFILE_LEGACY);
    int fusa = (header_type == ORA_HEADER_FUSA) && !legacy;

    uint64_t chan = 0;
    int found = 0;
    if (profile >= ORA_PROFILE_CUSTOM_BASE && profile < ORA_PROFILE_CUSTOM_BASE + g_n_custom) {
        const custom_profile* c = &g_custom[profile - ORA_PROFILE_CUSTOM_BASE];
        chan = c->chan_data_size;
        pf->n_fields = c->n_fields;
        memcpy(pf->fields, c->fields, sizeof(ora_field) * (size_t)c->n_fields);
        found = 1;
    } else {
        for (size_t i = 0; i < sizeof(PROFILES) / sizeof(PROFILES[0]); ++i) {
            if (PROFILES[i].profile != profile) continue;
            chan = PROFILES[i].chan_data_size;
            pf->n_fields = PROFILES[i].n;
            for (int k = 0; k < PROFILES[i].n; ++k) {
                const field_spec* s = &PROFILES[i].specs[k];
                strncpy(pf->fields[k].name, s->name, ORA_NAME_LEN - 1);
                if (ora_field_info(s->bit_start, s->bit_size, s->upshift, 0,
                                   s->num_elements, &pf->fields[k].info))
                    return -1;
            }
            found = 1;
            break;
        }
    }
    if (!found) return -1; /* "Unknown lidar udp profile" */
    qsort(pf->fields, (size_t)pf->n_fields, sizeof(ora_field), cmp_field_name);

    pf->profile = profile;
    pf->header_type = header_type;
    pf->pixels_per_column = h;
    pf->columns_per_packet = cpp;
    pf->columns_per_frame = w;
    pf->packet_header_size = legacy ? 0 : 32;
    pf->col_header_size = legacy ? 16 : 12;
    pf->channel_data_size = chan;
    pf->col_footer_size = legacy ? 4 : 0;
    pf->packet_footer_size = legacy ? 0 : 32;
    pf->col_size = pf->col_header_size + (uint64_t)h * chan + pf->col_footer_size;
    pf->lidar_packet_size =
        pf->packet_header_size + (uint64_t)cpp * pf->col_size + pf->packet_footer_size;
    if (pf->lidar_packet_size > 65535) return -3;
    pf->max_frame_id = fusa ? 0xffffffffu : 0xffffu;

    if (legacy) { /* parsing.cpp:479-510 */
        ora_field_info(0, 0, 0, 0, 1, &pf->packet_type_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->init_id_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->prod_sn_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->alert_flags_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->countdown_thermal_shutdown_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->countdown_shot_limiting_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->thermal_shutdown_info);
        ora_field_info(0, 0, 0, 0, 1, &pf->shot_limiting_info);
        ora_field_info(80, 16, 0, 0, 1, &pf->frame_id_info);
        uint64_t start_bit = 8 * (pf->col_size - pf->col_footer_size);
        ora_field_info(start_bit, 32, 0, (start_bit + 32) / 8, 1, &pf->col_status_info);
    } else if (fusa) { /* :511-522 */
        ora_field_info(0, 8, 0, 0, 1, &pf->packet_type_info);
        ora_field_info(32, 32, 0, 0, 1, &pf->frame_id_info);
        ora_field_info(8, 24, 0, 0, 1, &pf->init_id_info);
        ora_field_info(64, 8, 0, 0, 1, &pf->alert_flags_info);
        ora_field_info(88, 40, 0, 0, 1, &pf->prod_sn_info);
        ora_field_info(128, 8, 0, 0, 1, &pf->countdown_thermal_shutdown_info);
        ora_field_info(136, 8, 0, 0, 1, &pf->countdown_shot_limiting_info);
        ora_field_info(144, 4, 0, 0, 1, &pf->thermal_shutdown_info);
        ora_field_info(152, 4, 0, 0, 1, &pf->shot_limiting_info);
        ora_field_info(80, 16, 0, 0, 1, &pf->col_status_info);
    } else { /* :523-535 */
        ora_field_info(0, 16, 0, 0, 1, &pf->packet_type_info);
        ora_field_info(16, 16, 0, 0, 1, &pf->frame_id_info);
        ora_field_info(32, 24, 0, 0, 1, &pf->init_id_info);
        ora_field_info(56, 40, 0, 0, 1, &pf->prod_sn_info);
        ora_field_info(96, 8, 0, 0, 1, &pf->alert_flags_info);
        ora_field_info(128, 8, 0, 0, 1, &pf->countdown_thermal_shutdown_info);
        ora_field_info(136, 8, 0, 0, 1, &pf->countdown_shot_limiting_info);
        ora_field_info(144, 4, 0, 0, 1, &pf->thermal_shutdown_info);
        ora_field_info(152, 4, 0, 0, 1, &pf->shot_limiting_info);
        ora_field_info(80, 16, 0, 0, 1, &pf->col_status_info);
    }
    ora_field_info(0, 64, 0, 0, 1, &pf->col_timestamp_info);      /* :537 */
    ora_field_info(64, 16, 0, 0, 1, &pf->col_measurement_id_info); /* :538 */
    return 0;
}

const ora_fdi* ora_pf_field(const ora_pf* pf, const char* name) {
    for (int i = 0; i < pf->n_fields; ++i)
        if (strcmp(pf->fields[i].name, name) == 0) return &pf->fields[i].info;
    return NULL;
}

int ora_default_planes(int profile, char names[][ORA_NAME_LEN], int32_t* types,
                       int32_t* n_extra, int max_n) {
    if (profile >= ORA_PROFILE_CUSTOM_BASE && profile < ORA_PROFILE_CUSTOM_BASE + g_n_custom) {
        const custom_profile* c = &g_custom[profile - ORA_PROFILE_CUSTOM_BASE];
        int n = c->n_fields < max_n ? c->n_fields : max_n;
        for (int i = 0; i < n; ++i) {
            strncpy(names[i], c->fields[i].name, ORA_NAME_LEN);
            types[i] = c->slot_types[i];
            n_extra[i] = 1;
        }
        return n;
    }
    for (size_t i = 0; i < sizeof(SLOTS) / sizeof(SLOTS[0]); ++i) {
        if (SLOTS[i].profile != profile) continue;
        int n = SLOTS[i].n < max_n ? SLOTS[i].n : max_n;
        for (int k = 0; k < n; ++k) {
            memset(names[k], 0, ORA_NAME_LEN);
            strncpy(names[k], SLOTS[i].s[k].name, ORA_NAME_LEN - 1);
            types[k] = SLOTS[i].s[k].ty;
            /* lookup_frame_fields: RGB becomes h x w x 3 (lidar_frame.cpp:248-253) */
            n_extra[k] = strcmp(SLOTS[i].s[k].name, "RGB") == 0 ? 3 : 1;
        }
        return n;
    }
    return -1;
}

/* block_parsable: parsing.cpp:958-966 */
int ora_block_parsable(const ora_pf* pf) {
    static const int dims[3] = {16, 8, 4};
    for (int i = 0; i < 3; ++i)
        if (pf->pixels_per_column % (uint32_t)dims[i] == 0 &&
            pf->columns_per_packet % (uint32_t)dims[i] == 0)
            return dims[i];
    return 0;
}

/* accessors: parsing.cpp:736-836 */
const uint8_t* ora_nth_col(const ora_pf* pf, size_t i, const uint8_t* lidar_buf) {
    return lidar_buf + pf->packet_header_size + i * pf->col_size;
}
static const uint8_t* nth_px(const ora_pf* pf, size_t px, const uint8_t* col_buf) {
    return col_buf + pf->col_header_size + px * pf->channel_data_size;
}
uint32_t ora_frame_id(const ora_pf* pf, const uint8_t* b) { return (uint32_t)ora_fdi_get(&pf->frame_id_info, b); }
uint32_t ora_init_id(const ora_pf* pf, const uint8_t* b) { return (uint32_t)ora_fdi_get(&pf->init_id_info, b); }
uint64_t ora_prod_sn(const ora_pf* pf, const uint8_t* b) { return ora_fdi_get(&pf->prod_sn_info, b); }
uint16_t ora_packet_type(const ora_pf* pf, const uint8_t* b) { return (uint16_t)ora_fdi_get(&pf->packet_type_info, b); }
uint8_t ora_alert_flags(const ora_pf* pf, const uint8_t* b) { return (uint8_t)ora_fdi_get(&pf->alert_flags_info, b); }
uint16_t ora_col_measurement_id(const ora_pf* pf, const uint8_t* c) { return (uint16_t)ora_fdi_get(&pf->col_measurement_id_info, c); }
uint64_t ora_col_timestamp(const ora_pf* pf, const uint8_t* c) { return ora_fdi_get(&pf->col_timestamp_info, c); }
uint32_t ora_col_status(const ora_pf* pf, const uint8_t* c) { return (uint32_t)ora_fdi_get(&pf->col_status_info, c); }
uint32_t ora_col_encoder(const ora_pf* pf, const uint8_t* c) { /* :807-815 */
    uint32_t r = 0;
    if (pf->profile == ORA_PROFILE_LEGACY) memcpy(&r, c + 12, 4);
    return r;
}
uint16_t ora_col_frame_id(const ora_pf* pf, const uint8_t* c) { /* :817-825 */
    uint16_t r = 0;
    if (pf->profile == ORA_PROFILE_LEGACY) memcpy(&r, c + 10, 2);
    return r;
}

/* frame_id_difference: parsing.cpp:1312-1321 */
int ora_frame_id_difference(const ora_pf* pf, uint32_t current, uint32_t other) {
    int64_t half = pf->max_frame_id >> 1;
    int64_t delta = (int64_t)other - (int64_t)current;
    if (delta < -half) delta += (int64_t)pf->max_frame_id + 1;
    else if (delta > half) delta -= (int64_t)pf->max_frame_id + 1;
    return (int)delta;
}

/* col_field<T>: parsing.cpp:659-675.  dst_stride is in elements of the
 * destination type, dst_elem_size = sizeof(T). */
int ora_col_field(const ora_pf* pf, const uint8_t* col_buf, const char* name,
                  void* dst, size_t dst_elem_size, int dst_stride) {
    const ora_fdi* f = ora_pf_field(pf, name);
    if (!f) return -1; /* std::map::at throws out_of_range */
    if (dst_elem_size < ora_type_size(f->ty_tag) * (size_t)f->num_elements)
        return -2; /* "Dest type too small for specified field" */
    uint8_t* d = (uint8_t*)dst;
    for (uint32_t px = 0; px < pf->pixels_per_column; ++px) {
        uint64_t word = ora_fdi_get(f, nth_px(pf, px, col_buf));
        memcpy(d + (size_t)px * (size_t)dst_stride * dst_elem_size, &word, dst_elem_size);
    }
    return 0;
}

/* block_field<T,BlockDim>: parsing.cpp:628-657.  The reference instantiates the loop per destination type T and block
 * size; the restatement does the same through a macro (a run-time element size in the innermost memcpy made this port 3.3x
 * slower than the reference's own loop, VERDICT r05 -- the arithmetic is unchanged). */
#define ORA_BLOCK_LOOP(T, BD)                                                                   \
    for (uint32_t px = 0; px < pf->pixels_per_column; ++px) {                                   \
        T* row = (T*)data + (ptrdiff_t)cols * px + m_id;                                        \
        for (int x = 0; x < (BD); ++x) row[x] = (T)ora_fdi_get(f, nth_px(pf, px, col_buf[x]));  \
    }
#define ORA_BLOCK_BY_DIM(T)                                   \
    switch (block_dim) {                                      \
        case 16: ORA_BLOCK_LOOP(T, 16) break;                 \
        case 8: ORA_BLOCK_LOOP(T, 8) break;                   \
        case 4: ORA_BLOCK_LOOP(T, 4) break;                   \
        default: ORA_BLOCK_LOOP(T, block_dim) break;          \
    }
int ora_block_field(const ora_pf* pf, void* data, size_t dst_elem_size,
                    int cols, const char* name, const uint8_t* lidar_buf,
                    int block_dim) {
    const ora_fdi* f = ora_pf_field(pf, name);
    if (!f) return -1;
    if (dst_elem_size < ora_type_size(f->ty_tag) * (size_t)f->num_elements)
        return -2;
    uint8_t* d = (uint8_t*)data;
    const uint8_t* col_buf[16];
    for (uint32_t icol = 0; icol < pf->columns_per_packet; icol += (uint32_t)block_dim) {
        for (int i = 0; i < block_dim; ++i)
            col_buf[i] = ora_nth_col(pf, icol + (uint32_t)i, lidar_buf);
        uint16_t m_id = ora_col_measurement_id(pf, col_buf[0]);
        switch (dst_elem_size) {
            case 1: ORA_BLOCK_BY_DIM(uint8_t) break;
            case 2: ORA_BLOCK_BY_DIM(uint16_t) break;
            case 4: ORA_BLOCK_BY_DIM(uint32_t) break;
            case 8: ORA_BLOCK_BY_DIM(uint64_t) break;
            default:   /* packed elements (3 x float16): little-endian truncation of the 64-bit word, as get<T> does */
                for (uint32_t px = 0; px < pf->pixels_per_column; ++px) {
                    ptrdiff_t f_offset = (ptrdiff_t)cols * px + m_id;
                    for (int x = 0; x < block_dim; ++x) {
                        uint64_t word = ora_fdi_get(f, nth_px(pf, px, col_buf[x]));
                        memcpy(d + (size_t)(f_offset + x) * dst_elem_size, &word, dst_elem_size);
                    }
                }
        }
    }
    return 0;
}
#undef ORA_BLOCK_BY_DIM
#undef ORA_BLOCK_LOOP

/* set_block<T>: parsing.cpp:1056-1090 */
int ora_set_block(const ora_pf* pf, const void* data, size_t elem_size,
                  int cols, const char* name, uint8_t* lidar_buf) {
    if (pf->columns_per_packet > 32) return -3;
    const ora_fdi* f = ora_pf_field(pf, name);
    if (!f) return -1;
    uint8_t* col_buf[32];
    int valid[32];
    for (uint32_t i = 0; i < pf->columns_per_packet; ++i) {
        col_buf[i] = (uint8_t*)ora_nth_col(pf, i, lidar_buf);
        valid[i] = (int)(ora_col_status(pf, col_buf[i]) & 0x01);
    }
    uint16_t m_id = ora_col_measurement_id(pf, col_buf[0]);
    const uint8_t* s = (const uint8_t*)data;
    for (uint32_t px = 0; px < pf->pixels_per_column; ++px) {
        ptrdiff_t f_offset = (ptrdiff_t)cols * px + m_id;
        for (uint32_t x = 0; x < pf->columns_per_packet; ++x) {
            if (!valid[x]) continue;
            uint64_t word = 0;
            memcpy(&word, s + (size_t)(f_offset + x) * elem_size, elem_size);
            ora_fdi_set(f, (uint8_t*)nth_px(pf, px, col_buf[x]), word);
        }
    }
    return 0;
}

/* CRC64 (ECMA-182, reflected): parsing.cpp:1183-1217 */
uint64_t ora_crc64(const uint8_t* buf, size_t len) {
    static uint64_t table[256];
    static int init = 0;
    if (!init) {
        const uint64_t poly = 0xC96C5795D7870F42ull;
        for (uint32_t i = 0; i < 256; ++i) {
            uint64_t r = i;
            for (int j = 0; j < 8; ++j) r = (r >> 1) ^ (poly & ~((r & 1) - 1));
            table[i] = r;
        }
        init = 1;
    }
    uint64_t crc = ~0ull;
    while (len--) crc = table[*buf++ ^ (crc & 0xFF)] ^ (crc >> 8);
    return ~crc;
}

/* ------------------------------------------------------------------------- */
/* LidarFrame: ouster_core/src/lidar_frame.cpp:327-359                       */
/* ------------------------------------------------------------------------- */
ora_frame* ora_frame_new(uint32_t h, uint32_t w, uint32_t cpp) {
    if ((uint64_t)w * h == 0 || cpp == 0) return NULL;
    ora_frame* f = (ora_frame*)calloc(1, sizeof *f);
    f->h = h; f->w = w; f->cpp = cpp;
    f->n_packets = (w + cpp - 1) / cpp;
    f->timestamp = (uint64_t*)calloc(w, 8);
    f->measurement_id = (uint16_t*)calloc(w, 2);
    f->status = (uint32_t*)calloc(w, 4);
    f->packet_timestamp = (uint64_t*)calloc(f->n_packets, 8);
    f->alert_flags = (uint8_t*)calloc(f->n_packets, 1);
    f->frame_id = -1;
    return f;
}

void ora_frame_free(ora_frame* f) {
    if (!f) return;
    for (int i = 0; i < f->n_planes; ++i) free(f->planes[i].data);
    free(f->timestamp); free(f->measurement_id); free(f->status);
    free(f->packet_timestamp); free(f->alert_flags);
    free(f);
}

static ora_plane* find_plane(const ora_frame* f, const char* name) {
    for (int i = 0; i < f->n_planes; ++i)
        if (strcmp(f->planes[i].name, name) == 0) return (ora_plane*)&f->planes[i];
    return NULL;
}

static size_t plane_bytes(const ora_frame* f, const ora_plane* p) {
    return (size_t)f->h * f->w * (size_t)p->n_extra * ora_type_size(p->ty_tag);
}

int ora_frame_add_plane(ora_frame* f, const char* name, int ty_tag, int n_extra) {
    if (f->n_planes >= ORA_MAX_FIELDS || find_plane(f, name)) return -1;
    ora_plane* p = &f->planes[f->n_planes];
    memset(p, 0, sizeof *p);
    strncpy(p->name, name, ORA_NAME_LEN - 1);
    p->ty_tag = ty_tag;
    p->n_extra = n_extra < 1 ? 1 : n_extra;
    p->data = calloc(1, plane_bytes(f, p)); /* Field is calloc'd: field.cpp:247-296 */
    f->n_planes++;
    return 0;
}

/* get_field_types(format, fw): lidar_frame.cpp:1038-1113 (WINDOW dropped for fw<3.2) */
int ora_frame_add_default_planes(ora_frame* f, int profile, int with_window) {
    char names[ORA_MAX_FIELDS][ORA_NAME_LEN];
    int32_t types[ORA_MAX_FIELDS], extra[ORA_MAX_FIELDS];
    int n = ora_default_planes(profile, names, types, extra, ORA_MAX_FIELDS);
    if (n < 0) return n;
    for (int i = 0; i < n; ++i) {
        if (!with_window && strcmp(names[i], "WINDOW") == 0) continue;
        if (ora_frame_add_plane(f, names[i], types[i], extra[i])) return -1;
    }
    return 0;
}

void* ora_frame_plane(ora_frame* f, const char* name) {
    ora_plane* p = find_plane(f, name);
    return p ? p->data : NULL;
}
int ora_frame_plane_type(const ora_frame* f, const char* name) {
    ora_plane* p = find_plane(f, name);
    return p ? p->ty_tag : -1;
}

void ora_frame_fill(ora_frame* f, int v) {
    for (int i = 0; i < f->n_planes; ++i) memset(f->planes[i].data, v, plane_bytes(f, &f->planes[i]));
    memset(f->timestamp, v, (size_t)f->w * 8);
    memset(f->measurement_id, v, (size_t)f->w * 2);
    memset(f->status, v, (size_t)f->w * 4);
    memset(f->packet_timestamp, v, (size_t)f->n_packets * 8);
    memset(f->alert_flags, v, f->n_packets);
}

/* ------------------------------------------------------------------------- */
/* FrameBatcher (lidar path): ouster_core/src/lidar_frame.cpp:1248-1959      */
/* ------------------------------------------------------------------------- */
#define CACHE_CAP 64
typedef struct { uint8_t* buf; size_t len; uint64_t ts; uint64_t seq; } cached_packet;

struct ora_batcher {
    ora_pf pf;
    size_t max_cache_size;           /* lidar_frame.h:999 (default 4) */
    uint16_t next_valid_m_id;        /* :1001 */
    cached_packet cache[CACHE_CAP];
    int n_cache;
    uint64_t seq;
    int64_t finished_frame_id;       /* :1007 */
    int64_t last_frame_id;
    int64_t last_init_id;
    int reset_frame;                 /* :1011 */
    size_t expected_lidar_packets;
    size_t batched_lidar_packets;
    size_t dropped_packets;
    int force_col_path;
};

ora_batcher* ora_batcher_new(const ora_pf* pf, int64_t init_id, uint32_t expected) {
    if (pf->columns_per_packet == 0 || pf->pixels_per_column == 0) return NULL;
    ora_batcher* b = (ora_batcher*)calloc(1, sizeof *b);
    b->pf = *pf;
    b->max_cache_size = 4;
    b->finished_frame_id = -1;
    b->last_frame_id = -1;
    b->last_init_id = init_id;
    b->reset_frame = 1;
    b->expected_lidar_packets = expected;
    return b;
}

static void cache_clear(ora_batcher* b) {
    for (int i = 0; i < b->n_cache; ++i) free(b->cache[i].buf);
    b->n_cache = 0;
}

void ora_batcher_free(ora_batcher* b) {
    if (!b) return;
    cache_clear(b);
    free(b);
}

void ora_batcher_force_col_path(ora_batcher* b, int on) { b->force_col_path = on; }
uint64_t ora_batcher_dropped(const ora_batcher* b) { return b->dropped_packets; }

/* reset(): lidar_frame.cpp:1929-1940 */
void ora_batcher_reset(ora_batcher* b) {
    b->reset_frame = 1;
    b->finished_frame_id = -1;
    b->next_valid_m_id = 0;
    b->batched_lidar_packets = 0;
    cache_clear(b);
}

/* zero_header_cols: lidar_frame.cpp:1274-1278 */
static void zero_header_cols(ora_frame* fr, ptrdiff_t start, ptrdiff_t end) {
    if (end <= start) return;
    memset(fr->timestamp + start, 0, (size_t)(end - start) * 8);
    memset(fr->measurement_id + start, 0, (size_t)(end - start) * 2);
    memset(fr->status + start, 0, (size_t)(end - start) * 4);
}

/* zero_field: lidar_frame.cpp:1371-1407 (float16 planes are "zeroed" to NaN 0x7e00) */
static void zero_plane(ora_frame* fr, ora_plane* p, ptrdiff_t start, ptrdiff_t end) {
    if (start == end) return;
    size_t es = ora_type_size(p->ty_tag) * (size_t)p->n_extra;
    size_t row = (size_t)fr->w * es;
    size_t slice = (size_t)(end - start) * es;
    uint8_t* base = (uint8_t*)p->data + (size_t)start * es;
    for (uint32_t r = 0; r < fr->h; ++r) {
        if (p->ty_tag == ORA_F16) {
            uint16_t* d = (uint16_t*)(base + r * row);
            for (size_t i = 0; i < slice / 2; ++i) d[i] = 0x7e00;
        } else {
            memset(base + r * row, 0, slice);
        }
    }
}

/* zero_fields: lidar_frame.cpp:1409-1418 -- only planes named in the packet format */
static void zero_fields(ora_frame* fr, const ora_pf* pf, ptrdiff_t start, ptrdiff_t end) {
    for (int i = 0; i < pf->n_fields; ++i) {
        ora_plane* p = find_plane(fr, pf->fields[i].name);
        if (p) zero_plane(fr, p, start, end);
    }
}

/* parse_by_col: lidar_frame.cpp:1422-1466 (RAW_HEADERS handling omitted: out of scope) */
static void parse_by_col(ora_batcher* b, const uint8_t* pkt, ora_frame* fr) {
    const ora_pf* pf = &b->pf;
    for (uint32_t icol = 0; icol < pf->columns_per_packet; ++icol) {
        const uint8_t* col = ora_nth_col(pf, icol, pkt);
        uint16_t m_id = ora_col_measurement_id(pf, col);
        uint64_t ts = ora_col_timestamp(pf, col);
        uint32_t status = ora_col_status(pf, col);
        int valid = (status & 0x01) != 0;
        if (m_id >= fr->w) continue;
        if (!valid) continue;
        if (m_id >= b->next_valid_m_id) {
            zero_fields(fr, pf, b->next_valid_m_id, m_id);
            zero_header_cols(fr, b->next_valid_m_id, m_id);
            b->next_valid_m_id = (uint16_t)(m_id + 1);
        }
        fr->timestamp[m_id] = ts;
        fr->measurement_id[m_id] = m_id;
        fr->status[m_id] = status;
        /* foreach_channel_field_ndim(... ParseFieldCol ...): lidar_frame_impl.h:367-375 */
        for (int i = 0; i < pf->n_fields; ++i) {
            ora_plane* p = find_plane(fr, pf->fields[i].name);
            if (!p) continue;
            size_t es = ora_type_size(p->ty_tag) * (size_t)p->n_extra;
            ora_col_field(pf, col, pf->fields[i].name,
                          (uint8_t*)p->data + (size_t)m_id * es, es, (int)fr->w);
        }
    }
}

/* parse_by_block: lidar_frame.cpp:1492-1528 */
static int parse_by_block(ora_batcher* b, const uint8_t* pkt, ora_frame* fr) {
    const ora_pf* pf = &b->pf;
    uint16_t first_m_id = ora_col_measurement_id(pf, ora_nth_col(pf, 0, pkt));
    if (first_m_id >= b->next_valid_m_id) {
        zero_fields(fr, pf, b->next_valid_m_id, first_m_id);
        zero_header_cols(fr, b->next_valid_m_id, first_m_id);
        b->next_valid_m_id = (uint16_t)(first_m_id + pf->columns_per_packet);
    }
    for (uint32_t icol = 0; icol < pf->columns_per_packet; ++icol) {
        const uint8_t* col = ora_nth_col(pf, icol, pkt);
        uint16_t m_id = ora_col_measurement_id(pf, col);
        fr->measurement_id[m_id] = m_id;
        fr->timestamp[m_id] = ora_col_timestamp(pf, col);
        fr->status[m_id] = ora_col_status(pf, col);
    }
    int bd = ora_block_parsable(pf);
    if (bd == 0) return -1;
    for (int i = 0; i < pf->n_fields; ++i) {
        ora_plane* p = find_plane(fr, pf->fields[i].name);
        if (!p) continue;
        size_t es = ora_type_size(p->ty_tag) * (size_t)p->n_extra;
        ora_block_field(pf, p->data, es, (int)fr->w, pf->fields[i].name, pkt, bd);
    }
    return 0;
}

/* batch_lidar_packet: lidar_frame.cpp:1530-1576 */
static void batch_lidar_packet(ora_batcher* b, const uint8_t* pkt, uint64_t host_ts,
                               ora_frame* fr) {
    const ora_pf* pf = &b->pf;
    const uint8_t* col0 = ora_nth_col(pf, 0, pkt);
    uint16_t packet_id = (uint16_t)(ora_col_measurement_id(pf, col0) / pf->columns_per_packet);
    if (packet_id < fr->n_packets) {
        fr->packet_timestamp[packet_id] = host_ts;
        fr->alert_flags[packet_id] = ora_alert_flags(pf, pkt);
    }
    size_t block_parsable = (size_t)ora_block_parsable(pf);
    for (uint32_t icol = 0; icol < pf->columns_per_packet; ++icol) {
        const uint8_t* col = ora_nth_col(pf, icol, pkt);
        uint16_t m_id = ora_col_measurement_id(pf, col);
        uint32_t status = ora_col_status(pf, col);
        if (!(status & 0x01) || m_id >= fr->w) { block_parsable = 0; break; }
    }
    if (block_parsable != 0) {
        for (uint32_t icol = 0; icol < pf->columns_per_packet; icol += (uint32_t)block_parsable) {
            uint16_t m_id = ora_col_measurement_id(pf, ora_nth_col(pf, icol, pkt));
            if (m_id + block_parsable > fr->w) { block_parsable = 0; break; }
        }
    }
    if (block_parsable != 0 && !b->force_col_path) parse_by_block(b, pkt, fr);
    else parse_by_col(b, pkt, fr);
    b->batched_lidar_packets++;
}

/* frame_status(): lidar_frame.cpp:1310-1323 */
static uint64_t make_frame_status(uint8_t thermal, uint8_t shot) {
    return (uint64_t)(thermal & 0x0f) | ((uint64_t)(shot & 0x0f) << 4);
}

/* start_frame: lidar_frame.cpp:1709-1741 */
static void start_frame(ora_batcher* b, int64_t f_id, const uint8_t* pkt, ora_frame* fr) {
    const ora_pf* pf = &b->pf;
    b->finished_frame_id = -1;
    b->next_valid_m_id = 0;
    b->batched_lidar_packets = 0;
    fr->frame_id = f_id;
    zero_header_cols(fr, 0, (ptrdiff_t)fr->w);
    memset(fr->packet_timestamp, 0, (size_t)fr->n_packets * 8);
    uint8_t th = (uint8_t)ora_fdi_get(&pf->thermal_shutdown_info, pkt);
    uint8_t sl = (uint8_t)ora_fdi_get(&pf->shot_limiting_info, pkt);
    fr->frame_status = make_frame_status(th, sl);
    fr->shutdown_countdown = (uint16_t)ora_fdi_get(&pf->countdown_thermal_shutdown_info, pkt);
    fr->shot_limiting_countdown = (uint16_t)ora_fdi_get(&pf->countdown_shot_limiting_info, pkt);
}

/* check_frame_complete: lidar_frame.cpp:1894-1903 (lidar only) */
static int frame_complete(const ora_batcher* b, const ora_frame* fr) {
    size_t nz = 0;
    for (uint32_t i = 0; i < fr->n_packets; ++i) nz += fr->packet_timestamp[i] != 0;
    return b->batched_lidar_packets >= b->expected_lidar_packets &&
           nz == b->expected_lidar_packets;
}

/* finalize_frame: lidar_frame.cpp:1905-1927; returns <0 for the FUSA frame-id
 * regression error (:1915-1918) */
static int finalize_frame(ora_batcher* b, ora_frame* fr, int64_t sensor_init_id) {
    if (b->next_valid_m_id < fr->w)
        zero_fields(fr, &b->pf, b->next_valid_m_id, (ptrdiff_t)fr->w);
    if (sensor_init_id == b->last_init_id && fr->frame_id <= b->last_frame_id &&
        b->pf.header_type == ORA_HEADER_FUSA)
        return -1;
    b->finished_frame_id = fr->frame_id;
    b->last_frame_id = fr->frame_id;
    b->batched_lidar_packets = 0;
    return 0;
}

static void cache_push(ora_batcher* b, const uint8_t* pkt, size_t len, uint64_t ts) {
    if (b->n_cache >= CACHE_CAP) return;
    cached_packet* c = &b->cache[b->n_cache++];
    c->buf = (uint8_t*)malloc(len + 8);
    memcpy(c->buf, pkt, len);
    memset(c->buf + len, 0, 8);
    c->len = len; c->ts = ts; c->seq = b->seq++;
}

/* top of the priority queue: lowest frame id (PacketComparator, lidar_frame.h:970-993);
 * ties resolved first-in-first-out */
static int cache_top(const ora_batcher* b) {
    int best = 0;
    for (int i = 1; i < b->n_cache; ++i) {
        int d = ora_frame_id_difference(&b->pf, ora_frame_id(&b->pf, b->cache[best].buf),
                                        ora_frame_id(&b->pf, b->cache[i].buf));
        if (d < 0 || (d == 0 && b->cache[i].seq < b->cache[best].seq)) best = i;
    }
    return best;
}
static void cache_pop(ora_batcher* b, int i) {
    free(b->cache[i].buf);
    b->cache[i] = b->cache[--b->n_cache];
}

/* batch_with_caching: lidar_frame.cpp:1743-1793 */
static int batch_with_caching(ora_batcher* b, const uint8_t* pkt, size_t len, uint64_t ts,
                              ora_frame* fr, int64_t sensor_init_id) {
    cache_push(b, pkt, len, ts);
    while (b->n_cache > 0) {
        int t = cache_top(b);
        const uint8_t* buf = b->cache[t].buf;
        int64_t f_id = ora_frame_id(&b->pf, buf);
        if (b->finished_frame_id >= 0 &&
            ora_frame_id_difference(&b->pf, (uint32_t)b->finished_frame_id, (uint32_t)f_id) <= 0) {
            b->dropped_packets++;
            cache_pop(b, t);
            continue;
        }
        if (fr->frame_id == -1 || b->finished_frame_id >= 0) start_frame(b, f_id, buf, fr);
        int diff = ora_frame_id_difference(&b->pf, (uint32_t)fr->frame_id, (uint32_t)f_id);
        if (diff < 0) {
            b->dropped_packets++;
            cache_pop(b, t);
        } else if (diff > 0) {
            if ((size_t)b->n_cache >= b->max_cache_size) {
                if (finalize_frame(b, fr, sensor_init_id)) return -1;
                return 1;
            }
            return 0;
        } else {
            batch_lidar_packet(b, buf, b->cache[t].ts, fr);
            cache_pop(b, t);
            if (frame_complete(b, fr)) {
                if (finalize_frame(b, fr, sensor_init_id)) return -1;
                return 1;
            }
        }
    }
    return 0;
}

/* Test helper: release the frame being assembled the way batch() does when it gives up
 * waiting (finalize_frame, lidar_frame.cpp:1905-1927): zero the tail past the last
 * received column.  The reference has no public entry for this; it happens inside
 * batch_with_caching / handle_init_id_change. */
int ora_batcher_finalize(ora_batcher* b, ora_frame* fr) {
    return finalize_frame(b, fr, -1);
}

/* batch(): lidar_frame.cpp:1824-1884; handle_init_id_change :1795-1822.
 * The frame's sensor_info->init_id is taken equal to the batcher's constructor
 * init id (single-sensor use). */
int ora_batcher_batch(ora_batcher* b, const uint8_t* pkt, size_t len,
                      uint64_t host_ts, ora_frame* fr) {
    const ora_pf* pf = &b->pf;
    static int64_t dummy;
    (void)dummy;
    int64_t sensor_init_id = b->last_init_id;
    if (b->reset_frame) { fr->frame_id = -1; b->reset_frame = 0; }
    if (fr->w != pf->columns_per_frame || fr->h != pf->pixels_per_column) return -2;
    if (fr->n_packets != fr->w / pf->columns_per_packet) return -3;

    if (pf->profile != ORA_PROFILE_LEGACY && (int64_t)ora_init_id(pf, pkt) != b->last_init_id) {
        b->last_init_id = ora_init_id(pf, pkt);
        if (fr->frame_id == -1 || b->finished_frame_id >= 0) {
            ora_batcher_reset(b);
            b->reset_frame = 0;
            start_frame(b, ora_frame_id(pf, pkt), pkt, fr);
            batch_lidar_packet(b, pkt, host_ts, fr);
            if (frame_complete(b, fr)) { finalize_frame(b, fr, -1); return 1; }
            return 0;
        }
        finalize_frame(b, fr, -1);
        ora_batcher_reset(b);
        cache_push(b, pkt, len, host_ts);
        return 1;
    }

    int64_t f_id = ora_frame_id(pf, pkt);
    if (b->n_cache == 0) {
        if (b->finished_frame_id >= 0 &&
            ora_frame_id_difference(pf, (uint32_t)b->finished_frame_id, (uint32_t)f_id) <= 0) {
            b->dropped_packets++;
            return 0;
        }
        if (fr->frame_id == -1 || b->finished_frame_id >= 0) {
            start_frame(b, f_id, pkt, fr);
            batch_lidar_packet(b, pkt, host_ts, fr);
            if (frame_complete(b, fr)) {
                if (finalize_frame(b, fr, sensor_init_id)) return -1;
                return 1;
            }
            return 0;
        }
    }
    if (fr->frame_id == f_id && b->finished_frame_id < 0) {
        batch_lidar_packet(b, pkt, host_ts, fr);
        if (frame_complete(b, fr)) {
            if (finalize_frame(b, fr, sensor_init_id)) return -1;
            return 1;
        }
        return 0;
    }
    return batch_with_caching(b, pkt, len, host_ts, fr, sensor_init_id);
}

/* ------------------------------------------------------------------------- */
/* frame_to_packets (lidar part): impl/lidar_frame_impl.h:435-531            */
/* ------------------------------------------------------------------------- */
int ora_frame_to_packets(const ora_frame* fr, const ora_pf* pf, uint32_t init_id,
                         uint64_t prod_sn, uint8_t* out, uint64_t* out_ts) {
    if (fr->w / pf->columns_per_packet != fr->n_packets) return -1;
    int emitted = 0;
    size_t psz = pf->lidar_packet_size;
    for (uint32_t packet_id = 0; packet_id < fr->n_packets; ++packet_id) {
        uint8_t* buf = out + (size_t)emitted * psz;
        memset(buf, 0, psz);
        uint64_t host_ts = fr->packet_timestamp[packet_id];
        /* set_header lambda :458-470 */
        ora_fdi_set(&pf->thermal_shutdown_info, buf, (uint8_t)(fr->frame_status & 0x0f));
        ora_fdi_set(&pf->shot_limiting_info, buf, (uint8_t)((fr->frame_status & 0xf0) >> 4));
        ora_fdi_set(&pf->countdown_thermal_shutdown_info, buf, (uint8_t)fr->shutdown_countdown);
        ora_fdi_set(&pf->countdown_shot_limiting_info, buf, (uint8_t)fr->shot_limiting_countdown);
        ora_fdi_set(&pf->frame_id_info, buf, (uint32_t)fr->frame_id);
        ora_fdi_set(&pf->init_id_info, buf, init_id);
        ora_fdi_set(&pf->prod_sn_info, buf, prod_sn);
        ora_fdi_set(&pf->packet_type_info, buf, 0x1);
        ora_fdi_set(&pf->alert_flags_info, buf, fr->alert_flags[packet_id]);

        int any_valid = 0;
        for (uint32_t icol = 0; icol < pf->columns_per_packet; ++icol) {
            uint8_t* col = (uint8_t*)ora_nth_col(pf, icol, buf);
            uint32_t id = packet_id * pf->columns_per_packet + icol;
            ora_fdi_set(&pf->col_status_info, col, fr->status[id]);
            ora_fdi_set(&pf->col_measurement_id_info, col, (uint16_t)id);
            ora_fdi_set(&pf->col_timestamp_info, col, fr->timestamp[id]);
            any_valid |= (int)(fr->status[id] & 0x01);
        }
        if (!any_valid && !host_ts) continue; /* :498-501 */

        for (int i = 0; i < pf->n_fields; ++i) { /* pack_field :503-514 */
            const ora_plane* p = find_plane(fr, pf->fields[i].name);
            if (!p) continue;
            size_t es = ora_type_size(p->ty_tag) * (size_t)p->n_extra;
            ora_set_block(pf, p->data, es, (int)fr->w, pf->fields[i].name, buf);
        }
        if (pf->profile != ORA_PROFILE_LEGACY && pf->header_type == ORA_HEADER_STANDARD) {
            uint64_t crc = ora_crc64(buf, psz - 8); /* :521-528 */
            memcpy(buf + psz - 8, &crc, 8);
        }
        if (out_ts) out_ts[emitted] = host_ts;
        emitted++;
    }
    return emitted;
}

/* ------------------------------------------------------------------------- */
/* destagger_into<T>: impl/lidar_frame_impl.h:733-760                        */
/* ------------------------------------------------------------------------- */
int ora_destagger(const void* img, void* out, size_t h, size_t w, size_t es,
                  const int32_t* shifts, size_t n_shifts, int inverse) {
    if (n_shifts != h) return -1; /* "image height does not match shifts size" */
    int sign = inverse ? -1 : +1;
    const uint8_t* g = (const uint8_t*)img;
    uint8_t* d = (uint8_t*)out;
    for (size_t u = 0; u < h; ++u) {
        const uint8_t* g_row = g + u * w * es;
        uint8_t* d_row = d + u * w * es;
        /* NOTE: `sign * shift` (int) is converted to size_t before `% w` --
         * exactly as the reference's expression evaluates */
        const int offset = (int)((w + (size_t)(sign * shifts[u]) % w) % w);
        memcpy(d_row, g_row + (w - (size_t)offset) * es, (size_t)offset * es);
        memcpy(d_row + (size_t)offset * es, g_row, (w - (size_t)offset) * es);
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* make_xyz_lut: ouster_core/src/xyzlut.cpp:11-89                            */
/* ------------------------------------------------------------------------- */
int ora_make_xyz_lut(size_t w, size_t h, double range_unit, const double* b2l,
                     const double* tf, const double* az_deg, const double* alt_deg,
                     size_t n_angles, double* direction, double* offset) {
    if (w == 0 || h == 0) return -1; /* "lut dimensions must be greater than zero" */
    if (n_angles != h && n_angles != w * h) return -2; /* "unexpected frame dimensions" */
    const double b2l_x = b2l[0 * 4 + 3], b2l_z = b2l[2 * 4 + 3];
    double n = b2l_x;
    if (b2l_z != 0) n = sqrt(pow(b2l_x, 2) + pow(b2l_z, 2));

    const double azimuth_radians = M_PI * 2.0 / (double)w;
    for (size_t col = 0; col < w; ++col) {
        for (size_t row = 0; row < h; ++row) {
            size_t i = row * w + col;
            double enc, azi, alt;
            if (n_angles == h) { /* OS sensor :33-47 */
                enc = 2.0 * M_PI - ((double)col * azimuth_radians);
                azi = -az_deg[row] * M_PI / 180.0;
                alt = alt_deg[row] * M_PI / 180.0;
            } else { /* DF sensor :49-59 */
                enc = 0;
                azi = az_deg[i] * M_PI / 180.0;
                alt = alt_deg[i] * M_PI / 180.0;
            }
            double dx = cos(enc + azi) * cos(alt);
            double dy = sin(enc + azi) * cos(alt);
            double dz = sin(alt);
            double ox = cos(enc) * b2l_x - dx * n;
            double oy = sin(enc) * b2l_x - dy * n;
            double oz = -dz * n + b2l_z;
            /* row-vector * R^T  ==  R * column-vector  (:78-82) */
            double d3[3] = {dx, dy, dz}, o3[3] = {ox, oy, oz};
            for (int r = 0; r < 3; ++r) {
                double dd = 0, oo = 0;
                for (int k = 0; k < 3; ++k) {
                    dd += d3[k] * tf[r * 4 + k];
                    oo += o3[k] * tf[r * 4 + k];
                }
                oo += tf[r * 4 + 3];
                direction[i * 3 + (size_t)r] = dd * range_unit;
                offset[i * 3 + (size_t)r] = oo * range_unit;
            }
        }
    }
    return 0;
}

/* cartesianT<T>: include/ouster/core/impl/cartesian.h:36-66 */
#define CARTESIAN_BODY(T)                                            \
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) {                   \
        const uint32_t r = range[i];                                 \
        if (r == 0) {                                                \
            pts[i * 3 + 0] = pts[i * 3 + 1] = pts[i * 3 + 2] = (T)0.0; \
        } else {                                                     \
            pts[i * 3 + 0] = r * dir[i * 3 + 0] + ofs[i * 3 + 0];    \
            pts[i * 3 + 1] = r * dir[i * 3 + 1] + ofs[i * 3 + 1];    \
            pts[i * 3 + 2] = r * dir[i * 3 + 2] + ofs[i * 3 + 2];    \
        }                                                            \
    }

void ora_cartesian_f64(double* pts, const uint32_t* range, const double* dir,
                       const double* ofs, size_t n) {
    CARTESIAN_BODY(double)
}
void ora_cartesian_f32(float* pts, const uint32_t* range, const float* dir,
                       const float* ofs, size_t n) {
    CARTESIAN_BODY(float)
}
void ora_cartesian_f64_omp(double* pts, const uint32_t* range, const double* dir,
                           const double* ofs, size_t n) {
#pragma omp parallel for schedule(static)
    CARTESIAN_BODY(double)
}
void ora_cartesian_f32_omp(float* pts, const uint32_t* range, const float* dir,
                           const float* ofs, size_t n) {
#pragma omp parallel for schedule(static)
    CARTESIAN_BODY(float)
}

/* dewarp<T>(dewarped, points, poses): include/ouster/core/pose_util.h:38-56.
 * points [h*w][3] row-major pixel order (ix = i*W + w), poses [W][16] (row-major 4x4 each),
 * arithmetic in T.  For T = float the poses are cast to float first, as MatrixX16R<float>. */
void ora_dewarp_f64(double* out, const double* pts, const double* poses, size_t h, size_t w) {
    for (size_t c = 0; c < w; ++c) {
        const double* m = poses + c * 16;
        for (size_t i = 0; i < h; ++i) {
            const size_t ix = i * w + c;
            const double x = pts[ix * 3], y = pts[ix * 3 + 1], z = pts[ix * 3 + 2];
            out[ix * 3 + 0] = m[0] * x + m[1] * y + m[2] * z + m[3];
            out[ix * 3 + 1] = m[4] * x + m[5] * y + m[6] * z + m[7];
            out[ix * 3 + 2] = m[8] * x + m[9] * y + m[10] * z + m[11];
        }
    }
}
void ora_dewarp_f32(float* out, const float* pts, const double* poses, size_t h, size_t w) {
    for (size_t c = 0; c < w; ++c) {
        float m[12];
        for (int k = 0; k < 12; ++k) m[k] = (float)poses[c * 16 + k];
        for (size_t i = 0; i < h; ++i) {
            const size_t ix = i * w + c;
            const float x = pts[ix * 3], y = pts[ix * 3 + 1], z = pts[ix * 3 + 2];
            out[ix * 3 + 0] = m[0] * x + m[1] * y + m[2] * z + m[3];
            out[ix * 3 + 1] = m[4] * x + m[5] * y + m[6] * z + m[7];
            out[ix * 3 + 2] = m[8] * x + m[9] * y + m[10] * z + m[11];
        }
    }
}

/* dewarp(LidarFrame, XYZLutT<T>, min_range, max_range) with provenance:
 * include/ouster/core/impl/dewarp_impl.h:23-81.  pts = xyzlut(range) (cartesianT<T> with the T LUT),
 * columns first_valid..last_valid (status & 1, lidar_frame.cpp:907-925), skipping status == 0,
 * rows top to bottom, keeping ceil(min*1e3) <= r <= floor(max*1e3); pt = R*pt + t in T with the
 * pose cast to T.  Returns the number of points written; col_idx / ts may be NULL. */
#define DEWARP_FRAME_BODY(T)                                                                     \
    int start = -1, stop = -1;                                                                   \
    for (size_t i = 0; i < w; ++i) if (status[i] & 1u) { start = (int)i; break; }                \
    for (int i = (int)w - 1; i >= 0; --i) if (status[i] & 1u) { stop = i; break; }               \
    if (start < 0 || stop < start) return 0;                                                     \
    const uint32_t min_r = (uint32_t)ceil(min_range * 1e3);                                      \
    const uint32_t max_r = (uint32_t)floor(max_range * 1e3);                                     \
    size_t n = 0;                                                                                \
    for (int x = start; x <= stop; ++x) {                                                        \
        if (status[x] == 0) continue;                                                            \
        T m[12];                                                                                 \
        for (int k = 0; k < 12; ++k) m[k] = (T)poses[(size_t)x * 16 + k];                        \
        for (size_t y = 0; y < h; ++y) {                                                         \
            const size_t ix = y * w + (size_t)x;                                                 \
            const uint32_t r = range[ix];                                                        \
            if (r < min_r || r > max_r) continue;                                                \
            T p[3];                                                                              \
            if (r == 0) { p[0] = p[1] = p[2] = (T)0; }                                           \
            else for (int k = 0; k < 3; ++k) p[k] = (T)r * dir[ix * 3 + k] + ofs[ix * 3 + k];    \
            out[n * 3 + 0] = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];                     \
            out[n * 3 + 1] = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];                     \
            out[n * 3 + 2] = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];                   \
            if (col_idx) col_idx[n] = (uint32_t)x;                                               \
            if (ts) ts[n] = timestamp[x];                                                        \
            ++n;                                                                                 \
        }                                                                                        \
    }                                                                                            \
    return n;

size_t ora_dewarp_frame_f64(double* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                            const uint32_t* status, const uint64_t* timestamp, const double* poses,
                            const double* dir, const double* ofs, size_t h, size_t w,
                            double min_range, double max_range) {
    DEWARP_FRAME_BODY(double)
}
size_t ora_dewarp_frame_f32(float* out, uint32_t* col_idx, uint64_t* ts, const uint32_t* range,
                            const uint32_t* status, const uint64_t* timestamp, const double* poses,
                            const float* dir, const float* ofs, size_t h, size_t w,
                            double min_range, double max_range) {
    DEWARP_FRAME_BODY(float)
}

/* ------------------------------------------------------------------------- */
/* CPU baseline driver: the reference's own sequence on a pool of frames      */
/* (cf. tests/benchmarks/core_benchmark.cpp:29-154)                          */
/* ------------------------------------------------------------------------- */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* flags: 1 = every thread works on ITS OWN copy of the LUT, the packet pool and the shifts, allocated and first touched by
 * that thread (NUMA-local pages; the shared 25 MB f64 LUT otherwise lives on the node of the thread that built it);
 * 2 = static schedule (frame f always on the same thread).  0 = the round-4 harness. */
double ora_bench_hot_path2(const ora_pf* pf, int with_window, const uint8_t* packets,
                           uint32_t pool_frames, uint32_t n_frames, uint32_t ppf, const int32_t* shifts,
                           const double* lut_dir_shared, const double* lut_ofs_shared, int xyz_f64,
                           int reps, int threads, uint64_t* checksum_out, int flags) {
    const uint32_t h = pf->pixels_per_column, w = pf->columns_per_frame;
    const size_t npx = (size_t)h * w;
    const size_t psz = pf->lidar_packet_size;
    float *fdir_shared = NULL, *fofs_shared = NULL;
    if (!xyz_f64) { /* XYZLutT<float>: cast of the double LUT, xyzlut.h:119-124 */
        fdir_shared = (float*)malloc(npx * 3 * sizeof(float));
        fofs_shared = (float*)malloc(npx * 3 * sizeof(float));
        for (size_t i = 0; i < npx * 3; ++i) { fdir_shared[i] = (float)lut_dir_shared[i]; fofs_shared[i] = (float)lut_ofs_shared[i]; }
    }
    const uint8_t* packets_shared = packets;
    if (threads < 1) threads = 1;
    uint64_t checksum = 0;
    double t0 = 0, t1 = 0;
#ifdef _OPENMP
    omp_set_num_threads(threads);
    omp_set_schedule((flags & 2) ? omp_sched_static : omp_sched_dynamic, (flags & 2) ? 0 : 1);
#endif
#pragma omp parallel reduction(+ : checksum)
    {
        ora_frame* fr = ora_frame_new(h, w, pf->columns_per_packet);
        ora_frame_add_default_planes(fr, pf->profile, with_window);
        ora_batcher* b = ora_batcher_new(pf, 0, ppf);
        static const char* dst_names[4] = {"RANGE", "RANGE2", "REFLECTIVITY", "REFLECTIVITY2"};
        void* dst[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            int ty = ora_frame_plane_type(fr, dst_names[k]);
            if (ty > 0) dst[k] = malloc(npx * ora_type_size(ty));
        }
        void* xyz1 = malloc(npx * 3 * (xyz_f64 ? 8 : 4));
        void* xyz2 = ora_frame_plane(fr, "RANGE2") ? malloc(npx * 3 * (xyz_f64 ? 8 : 4)) : NULL;
        const double *lut_dir = lut_dir_shared, *lut_ofs = lut_ofs_shared;
        const float *fdir = fdir_shared, *fofs = fofs_shared;
        const uint8_t* packets = packets_shared;
        void *own[3] = {0, 0, 0};
        if (flags & 1) {   /* this thread's own, first-touched inputs */
            const size_t es = xyz_f64 ? 8 : 4, lb = npx * 3 * es, pb = (size_t)pool_frames * ppf * psz;
            own[0] = malloc(lb); own[1] = malloc(lb); own[2] = malloc(pb);
            memcpy(own[0], xyz_f64 ? (const void*)lut_dir_shared : (const void*)fdir_shared, lb);
            memcpy(own[1], xyz_f64 ? (const void*)lut_ofs_shared : (const void*)fofs_shared, lb);
            memcpy(own[2], packets_shared, pb);
            if (xyz_f64) { lut_dir = (const double*)own[0]; lut_ofs = (const double*)own[1]; }
            else { fdir = (const float*)own[0]; fofs = (const float*)own[1]; }
            packets = (const uint8_t*)own[2];
        }
#pragma omp barrier
#pragma omp master
        t0 = now_s();
        for (int rep = 0; rep < reps; ++rep) {
#pragma omp for schedule(runtime)
            for (uint32_t f = 0; f < n_frames; ++f) {
                ora_batcher_reset(b);
                const uint8_t* fpk = packets + (size_t)(f % pool_frames) * ppf * psz;
                b->last_init_id = ora_init_id(pf, fpk);
                b->last_frame_id = -1;
                for (uint32_t p = 0; p < ppf; ++p)
                    ora_batcher_batch(b, fpk + (size_t)p * psz, psz, 1 + p, fr);
                for (int k = 0; k < 4; ++k)
                    if (dst[k])
                        ora_destagger(ora_frame_plane(fr, dst_names[k]), dst[k], h, w,
                                      ora_type_size(ora_frame_plane_type(fr, dst_names[k])),
                                      shifts, h, 0);
                const uint32_t* r1 = (const uint32_t*)ora_frame_plane(fr, "RANGE");
                const uint32_t* r2 = (const uint32_t*)ora_frame_plane(fr, "RANGE2");
                if (xyz_f64) {
                    ora_cartesian_f64((double*)xyz1, r1, lut_dir, lut_ofs, npx);
                    if (r2) ora_cartesian_f64((double*)xyz2, r2, lut_dir, lut_ofs, npx);
                    checksum += (uint64_t)(int64_t)(((double*)xyz1)[(f * 7919u) % (npx * 3)] * 1e6);
                } else {
                    ora_cartesian_f32((float*)xyz1, r1, fdir, fofs, npx);
                    if (r2) ora_cartesian_f32((float*)xyz2, r2, fdir, fofs, npx);
                    checksum += (uint64_t)(int64_t)(((float*)xyz1)[(f * 7919u) % (npx * 3)] * 1e6f);
                }
                if (dst[0]) checksum += ((uint32_t*)dst[0])[(f * 104729u) % npx];
            }
        }
#pragma omp barrier
#pragma omp master
        t1 = now_s();
        for (int k = 0; k < 4; ++k) free(dst[k]);
        free(xyz1); free(xyz2);
        free(own[0]); free(own[1]); free(own[2]);
        ora_batcher_free(b);
        ora_frame_free(fr);
    }
    free(fdir_shared); free(fofs_shared);
    if (checksum_out) *checksum_out = checksum;
    return t1 - t0;
}

double ora_bench_hot_path(const ora_pf* pf, int with_window, const uint8_t* packets,
                          uint32_t pool_frames, uint32_t n_frames, uint32_t ppf, const int32_t* shifts,
                          const double* lut_dir, const double* lut_ofs, int xyz_f64,
                          int reps, int threads, uint64_t* checksum_out) {
    return ora_bench_hot_path2(pf, with_window, packets, pool_frames, n_frames, ppf, shifts, lut_dir, lut_ofs, xyz_f64, reps,
                               threads, checksum_out, 0);
}

/* STREAM-style copy on the same cores: every thread copies its own first-touched `bytes_per_thread` array `reps` times.
 * Returns seconds; the caller quotes 2 * bytes * threads * reps / seconds next to the hot path's byte rate. */
double ora_bench_stream_copy(size_t bytes_per_thread, int reps, int threads) {
    double t0 = 0, t1 = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
    omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        uint8_t* a = (uint8_t*)malloc(bytes_per_thread);
        uint8_t* b = (uint8_t*)malloc(bytes_per_thread);
        memset(a, 1, bytes_per_thread);
        memset(b, 2, bytes_per_thread);
#pragma omp barrier
#pragma omp master
        t0 = now_s();
        for (int r = 0; r < reps; ++r) {
            memcpy(b, a, bytes_per_thread);
            __asm__ volatile("" : : "r"(b) : "memory");
        }
#pragma omp barrier
#pragma omp master
        t1 = now_s();
        free(a); free(b);
    }
    return t1 - t0;
}
