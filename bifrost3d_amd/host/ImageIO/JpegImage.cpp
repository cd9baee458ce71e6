// JpegImage.cpp -- see JpegImage.h. A JPEG (ITU-T T.81) decoder for the texture loaders: baseline and extended sequential
// Huffman (SOF0 / SOF1, 8 bit) and progressive (SOF2) scans, restart intervals, 1, 3 or 4 (Adobe CMYK / YCCK) components with any of the usual chroma
// subsamplings, JFIF YCbCr and Adobe RGB. The container and entropy decoding follow the standard; the three stages whose
// arithmetic the standard leaves to the implementation -- inverse DCT, chroma upsampling, YCbCr -> RGB -- use the integer
// arithmetic of the decoder the reference loads its textures with (stb_image 2.29 inside extensions/StbImageLoader): the
// slow-but-accurate integer IDCT of the IJG library at 12 fractional bits with 2 guard bits between the passes, the 3:1
// "triangle" chroma interpolation, and the 20 bit fixed point colour matrix. Decoded pixels are therefore the bytes that
// reference produces (tests/test_image_codecs_cpu.py compares against that decoder built from the reference tree).
#include "JpegImage.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

using namespace Bifrost::Assets;

namespace JpegImage {

namespace {

const uint8_t ZIGZAG[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};   // tail: runs past the block land on the last coefficient

struct HuffmanTable {
    // canonical code (T.81 annex C): codes of length l are consecutive, starting at first_code[l]
    uint8_t values[256];
    int32_t first_code[18], first_index[17], count[17];
    bool defined = false;
    void build(const uint8_t counts[16], const uint8_t* symbols) {
        int code = 0, index = 0;
        for (int l = 1; l <= 16; ++l) {
            first_code[l] = code; first_index[l] = index; count[l] = counts[l - 1];
            code = (code + counts[l - 1]) << 1;
            index += counts[l - 1];
        }
        std::memcpy(values, symbols, size_t(index));
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_predictor = 0;
    int blocks_w = 0, blocks_h = 0;       // blocks in the interleaved (MCU padded) layout
    int plane_w = 0, plane_h = 0;         // blocks_w * 8, blocks_h * 8
    int x = 0, y = 0;                     // component size in samples
    std::vector<uint8_t> plane;
    std::vector<int16_t> coefficients;    // progressive only: 64 per block
};

struct Decoder {
    const uint8_t* data; size_t size, pos = 0;
    // entropy-coded segment bit reader
    uint32_t bit_buffer = 0; int bit_count = 0; int marker = -1; bool hit_marker = false;
    uint16_t quantization[4][64];
    HuffmanTable dc_tables[4], ac_tables[4];
    Component components[4];
    int component_count = 0, width = 0, height = 0, h_max = 1, v_max = 1, mcu_w = 0, mcu_h = 0, mcus_x = 0, mcus_y = 0;
    bool progressive = false, jfif = false, frame_seen = false;
    int adobe_transform = -1;
    int restart_interval = 0;
    // scan state
    int scan_components = 0, scan_order[4], spectral_start = 0, spectral_end = 63, successive_high = 0, successive_low = 0, eob_run = 0;
    const char* error = nullptr;

    bool fail(const char* message) { if (!error) error = message; return false; }
    int read_u8() { return pos < size ? data[pos++] : 0; }
    int read_u16() { int hi = read_u8(); return (hi << 8) | read_u8(); }

    // ---- bit reader over an entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment (zeros are fed on) ----
    void reset_bits() { bit_buffer = 0; bit_count = 0; hit_marker = false; marker = -1; }
    void fill_bits() {
        while (bit_count <= 24) {
            int byte = 0;
            if (!hit_marker && pos < size) {
                byte = data[pos++];
                if (byte == 0xFF) {
                    int next = pos < size ? data[pos++] : 0xD9;
                    while (next == 0xFF && pos < size) next = data[pos++];
                    if (next != 0) { marker = next; hit_marker = true; byte = 0; }
                }
            }
            bit_buffer |= uint32_t(byte) << (24 - bit_count);
            bit_count += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bit_count < n) fill_bits();
        const int value = int(bit_buffer >> (32 - n));
        bit_buffer <<= n; bit_count -= n;
        return value;
    }
    int get_bit() { return get_bits(1); }
    // n magnitude bits as a signed value (T.81 F.2.2.1 EXTEND)
    int receive_extend(int n) {
        if (n == 0) return 0;
        const int v = get_bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode_symbol(const HuffmanTable& table) {
        if (bit_count < 16) fill_bits();
        int code = 0;
        for (int l = 1; l <= 16; ++l) {
            code = (code << 1) | int((bit_buffer >> (32 - l)) & 1u);
            const int offset = code - table.first_code[l];
            if (offset >= 0 && offset < table.count[l]) {
                bit_buffer <<= l; bit_count -= l;
                return table.values[table.first_index[l] + offset];
            }
        }
        fail("bad huffman code");
        return -1;
    }

    // ---- marker segments -----------------------------------------------------------------------------------------------------------
    bool read_quantization_tables(int length) {
        while (length > 0) {
            const int pq_tq = read_u8(), precision = pq_tq >> 4, index = pq_tq & 15;
            if (precision > 1 || index > 3) return fail("bad DQT");
            for (int i = 0; i < 64; ++i) quantization[index][ZIGZAG[i]] = uint16_t(precision ? read_u16() : read_u8());
            length -= precision ? 129 : 65;
        }
        return length == 0 || fail("bad DQT length");
    }
    bool read_huffman_tables(int length) {
        while (length > 0) {
            const int tc_th = read_u8(), table_class = tc_th >> 4, index = tc_th & 15;
            if (table_class > 1 || index > 3) return fail("bad DHT");
            uint8_t counts[16], symbols[256];
            int total = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = uint8_t(read_u8()); total += counts[i]; }
            if (total > 256) return fail("bad DHT counts");
            for (int i = 0; i < total; ++i) symbols[i] = uint8_t(read_u8());
            (table_class ? ac_tables : dc_tables)[index].build(counts, symbols);
            length -= 17 + total;
        }
        return length == 0 || fail("bad DHT length");
    }
    bool read_frame(int length, bool is_progressive) {
        if (frame_seen) return fail("more than one frame");
        if (read_u8() != 8) return fail("only 8 bit samples are supported");
        height = read_u16(); width = read_u16();
        component_count = read_u8();
        if (width == 0 || height == 0) return fail("empty image");
        if (component_count != 1 && component_count != 3 && component_count != 4) return fail("only 1, 3 or 4 components are supported");
        if (length != 8 + 3 * component_count) return fail("bad SOF length");
        progressive = is_progressive;
        h_max = v_max = 1;
        for (int c = 0; c < component_count; ++c) {
            Component& comp = components[c];
            comp.id = read_u8();
            const int hv = read_u8();
            comp.h = hv >> 4; comp.v = hv & 15; comp.tq = read_u8();
            if (comp.h < 1 || comp.h > 4 || comp.v < 1 || comp.v > 4 || comp.tq > 3) return fail("bad component");
            h_max = comp.h > h_max ? comp.h : h_max; v_max = comp.v > v_max ? comp.v : v_max;
        }
        for (int c = 0; c < component_count; ++c)
            if (h_max % components[c].h || v_max % components[c].v) return fail("fractional sampling ratios are not supported");
        mcu_w = 8 * h_max; mcu_h = 8 * v_max;
        mcus_x = (width + mcu_w - 1) / mcu_w; mcus_y = (height + mcu_h - 1) / mcu_h;
        // Planes are sized from the header: a frame that claims more blocks than the file has bits (every block costs a scan at least one) is refused
        // before anything is allocated.
        if (uint64_t(mcus_x) * uint64_t(mcus_y) > uint64_t(size) * 8u) return fail("frame dimensions exceed what the file can hold");
        for (int c = 0; c < component_count; ++c) {
            Component& comp = components[c];
            comp.x = (width * comp.h + h_max - 1) / h_max; comp.y = (height * comp.v + v_max - 1) / v_max;
            comp.blocks_w = mcus_x * comp.h; comp.blocks_h = mcus_y * comp.v;
            comp.plane_w = comp.blocks_w * 8; comp.plane_h = comp.blocks_h * 8;
            comp.plane.assign(size_t(comp.plane_w) * comp.plane_h, 0);
            if (progressive) comp.coefficients.assign(size_t(comp.blocks_w) * comp.blocks_h * 64, 0);
        }
        frame_seen = true;
        return true;
    }
    bool read_scan_header(int length) {
        scan_components = read_u8();
        if (scan_components < 1 || scan_components > component_count || length != 6 + 2 * scan_components) return fail("bad SOS");
        for (int i = 0; i < scan_components; ++i) {
            const int id = read_u8(), tables = read_u8();
            int which = -1;
            for (int c = 0; c < component_count; ++c) if (components[c].id == id) which = c;
            if (which < 0) return fail("SOS names an unknown component");
            components[which].td = tables >> 4; components[which].ta = tables & 15;
            if (components[which].td > 3 || components[which].ta > 3) return fail("bad table selector");
            scan_order[i] = which;
        }
        spectral_start = read_u8(); spectral_end = read_u8();
        const int approximation = read_u8();
        successive_high = approximation >> 4; successive_low = approximation & 15;
        if (progressive) {
            if (spectral_start > 63 || spectral_end > 63 || spectral_start > spectral_end || successive_high > 13 || successive_low > 13) return fail("bad progressive scan parameters");
            if (spectral_start == 0 && spectral_end != 0) return fail("a DC scan cannot carry AC coefficients");
            if (spectral_start != 0 && scan_components != 1) return fail("AC scans are not interleaved");
        } else { spectral_start = 0; spectral_end = 63; successive_high = successive_low = 0; }
        return true;
    }

    // ---- block decoding --------------------------------------------------------------------------------------------------------------
    bool decode_block_sequential(Component& comp, int16_t block[64]) {
        std::memset(block, 0, 128);
        const HuffmanTable &dc = dc_tables[comp.td], &ac = ac_tables[comp.ta];
        if (!dc.defined || !ac.defined) return fail("scan uses an undefined huffman table");
        const uint16_t* q = quantization[comp.tq];
        const int t = decode_symbol(dc);
        if (t < 0 || t > 15) return fail("bad DC code");
        comp.dc_predictor += receive_extend(t);
        if (comp.dc_predictor < -(1 << 20) || comp.dc_predictor > (1 << 20)) return fail("DC predictor out of range");      // corrupt stream: no signed overflow below
        block[0] = int16_t(int64_t(comp.dc_predictor) * q[0]);
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(ac);
            if (rs < 0) return false;
            const int run = rs >> 4, magnitude_bits = rs & 15;
            if (magnitude_bits == 0) {
                if (rs != 0xF0) break;      // end of block
                k += 16;
            } else {
                k += run;
                const int position = ZIGZAG[k++];
                block[position] = int16_t(receive_extend(magnitude_bits) * q[position]);
            }
        }
        return true;
    }
    bool decode_block_progressive_dc(Component& comp, int16_t* block) {
        if (successive_high == 0) {     // first pass: the DC difference, scaled by the point transform
            const HuffmanTable& dc = dc_tables[comp.td];
            if (!dc.defined) return fail("scan uses an undefined huffman table");
            const int t = decode_symbol(dc);
            if (t < 0 || t > 15) return fail("bad DC code");
            comp.dc_predictor += receive_extend(t);
            if (comp.dc_predictor < -(1 << 20) || comp.dc_predictor > (1 << 20)) return fail("DC predictor out of range");
            block[0] = int16_t(int64_t(comp.dc_predictor) * (1 << successive_low));
        } else if (get_bit())           // refinement: one more bit of precision
            block[0] = int16_t(block[0] + (1 << successive_low));
        return true;
    }
    bool decode_block_progressive_ac(Component& comp, int16_t* block) {
        const HuffmanTable& ac = ac_tables[comp.ta];
        if (!ac.defined) return fail("scan uses an undefined huffman table");
        if (successive_high == 0) {     // first pass over the band (T.81 G.1.2.2)
            if (eob_run) { --eob_run; return true; }
            for (int k = spectral_start; k <= spectral_end;) {
                const int rs = decode_symbol(ac);
                if (rs < 0) return false;
                const int run = rs >> 4, magnitude_bits = rs & 15;
                if (magnitude_bits == 0) {
                    if (run < 15) {
                        eob_run = 1 << run;
                        if (run) eob_run += get_bits(run);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += run;
                    const int position = ZIGZAG[k++];
                    block[position] = int16_t(receive_extend(magnitude_bits) * (1 << successive_low));
                }
            }
            return true;
        }
        // refinement pass (T.81 G.1.2.3): correction bits for the coefficients already non-zero, new +-1 coefficients in between
        const int16_t bit = int16_t(1 << successive_low);
        auto refine = [&](int16_t& coefficient) {
            if (coefficient != 0 && get_bit() && (coefficient & bit) == 0) coefficient = int16_t(coefficient > 0 ? coefficient + bit : coefficient - bit);
        };
        if (eob_run) {
            --eob_run;
            for (int k = spectral_start; k <= spectral_end; ++k) refine(block[ZIGZAG[k]]);
            return true;
        }
        int k = spectral_start;
        do {
            const int rs = decode_symbol(ac);
            if (rs < 0) return false;
            int run = rs >> 4, magnitude_bits = rs & 15, value = 0;
            if (magnitude_bits == 0) {
                if (run < 15) {
                    eob_run = (1 << run) - 1;
                    if (run) eob_run += get_bits(run);
                    run = 64;       // to the end of the band: only corrections remain
                }
            } else {
                if (magnitude_bits != 1) return fail("bad refinement code");
                value = get_bit() ? bit : -bit;
            }
            while (k <= spectral_end) {
                int16_t& coefficient = block[ZIGZAG[k++]];
                if (coefficient != 0) refine(coefficient);
                else {
                    if (run == 0) { coefficient = int16_t(value); break; }
                    --run;
                }
            }
        } while (k <= spectral_end);
        return true;
    }

    // ---- inverse DCT: IJG "islow" at 12 fractional bits, 2 guard bits kept between the column and the row pass -------------------------
    static int fixed(float x) { return int(double(x) * 4096 + 0.5); }     // truncation towards zero, also for the negative constants
    struct Idct1D { int even[4], odd[4]; };
    static Idct1D idct_1d(int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7) {
        static const int C_0_541 = fixed(0.5411961f), C_M1_848 = fixed(-1.847759065f), C_0_765 = fixed(0.765366865f), C_1_176 = fixed(1.175875602f), C_0_299 = fixed(0.298631336f),
                         C_2_053 = fixed(2.053119869f), C_3_073 = fixed(3.072711026f), C_1_501 = fixed(1.501321110f), C_M0_900 = fixed(-0.899976223f), C_M2_563 = fixed(-2.562915447f),
                         C_M1_962 = fixed(-1.961570560f), C_M0_390 = fixed(-0.390180644f);
        Idct1D r;
        // even part: rotation of (s2, s6), butterflies with s0 +- s4
        const int z = (s2 + s6) * C_0_541;
        const int e2 = z + s6 * C_M1_848, e3 = z + s2 * C_0_765;
        const int e0 = (s0 + s4) * 4096, e1 = (s0 - s4) * 4096;
        r.even[0] = e0 + e3; r.even[3] = e0 - e3; r.even[1] = e1 + e2; r.even[2] = e1 - e2;
        // odd part
        const int a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
        const int z5 = (a + b) * C_1_176;
        const int pa = a * C_M1_962, pb = b * C_M0_390, pc = z5 + c * C_M0_900, pd = z5 + d * C_M2_563;
        r.odd[0] = s7 * C_0_299 + pc + pa;     // pairs with even[3]
        r.odd[1] = s5 * C_2_053 + pd + pb;     // pairs with even[2]
        r.odd[2] = s3 * C_3_073 + pd + pa;     // pairs with even[1]
        r.odd[3] = s1 * C_1_501 + pc + pb;     // pairs with even[0]
        return r;
    }
    static uint8_t clamp_u8(int x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void inverse_dct(const int16_t block[64], uint8_t* out, int stride) {
        int columns[64];
        for (int i = 0; i < 8; ++i) {
            const int16_t* d = block + i;
            if (!(d[8] | d[16] | d[24] | d[32] | d[40] | d[48] | d[56])) {     // DC only column: constant
                const int dc = d[0] * 4;
                for (int k = 0; k < 8; ++k) columns[i + 8 * k] = dc;
                continue;
            }
            const Idct1D r = idct_1d(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
            for (int k = 0; k < 4; ++k) {
                const int e = r.even[k] + 512, o = r.odd[3 - k];
                columns[i + 8 * k] = (e + o) >> 10;
                columns[i + 8 * (7 - k)] = (e - o) >> 10;
            }
        }
        for (int i = 0; i < 8; ++i, out += stride) {
            const int* v = columns + 8 * i;
            const Idct1D r = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            for (int k = 0; k < 4; ++k) {
                const int e = r.even[k] + 65536 + (128 << 17), o = r.odd[3 - k];      // rounding + the level shift
                out[k] = clamp_u8((e + o) >> 17);
                out[7 - k] = clamp_u8((e - o) >> 17);
            }
        }
    }

    // ---- one scan ----------------------------------------------------------------------------------------------------------------------
    // After each MCU: at the end of a restart interval the segment must end at a RSTn marker (byte align, reset the predictors and
    // the end-of-band run); anything else there ends the scan. Returns false when the scan is over.
    bool next_interval(int& countdown) {
        if (restart_interval == 0 || --countdown > 0) return true;
        if (bit_count < 24) fill_bits();
        if (marker < 0xD0 || marker > 0xD7) return false;
        reset_bits();
        for (Component& c : components) c.dc_predictor = 0;
        eob_run = 0;
        countdown = restart_interval;
        return true;
    }
    bool decode_scan() {
        reset_bits();
        for (Component& c : components) c.dc_predictor = 0;
        eob_run = 0;
        int countdown = restart_interval ? restart_interval : 0x7FFFFFFF;
        int16_t block[64];
        if (scan_components == 1) {     // non-interleaved: the component's own blocks, only those that cover the image
            Component& comp = components[scan_order[0]];
            const int w = (comp.x + 7) >> 3, h = (comp.y + 7) >> 3;
            for (int by = 0; by < h; ++by)
                for (int bx = 0; bx < w; ++bx) {
                    if (progressive) {
                        int16_t* coefficients = comp.coefficients.data() + 64 * (size_t(by) * comp.blocks_w + bx);
                        if (!(spectral_start == 0 ? decode_block_progressive_dc(comp, coefficients) : decode_block_progressive_ac(comp, coefficients))) return false;
                    } else {
                        if (!decode_block_sequential(comp, block)) return false;
                        inverse_dct(block, comp.plane.data() + size_t(by) * 8 * comp.plane_w + size_t(bx) * 8, comp.plane_w);
                    }
                    if (!next_interval(countdown)) return true;
                }
            return true;
        }
        for (int my = 0; my < mcus_y; ++my)
            for (int mx = 0; mx < mcus_x; ++mx) {
                for (int s = 0; s < scan_components; ++s) {
                    Component& comp = components[scan_order[s]];
                    for (int v = 0; v < comp.v; ++v)
                        for (int h = 0; h < comp.h; ++h) {
                            const int bx = mx * comp.h + h, by = my * comp.v + v;
                            if (progressive) {
                                if (!decode_block_progressive_dc(comp, comp.coefficients.data() + 64 * (size_t(by) * comp.blocks_w + bx))) return false;
                            } else {
                                if (!decode_block_sequential(comp, block)) return false;
                                inverse_dct(block, comp.plane.data() + size_t(by) * 8 * comp.plane_w + size_t(bx) * 8, comp.plane_w);
                            }
                        }
                }
                if (!next_interval(countdown)) return true;
            }
        return true;
    }
    void finish_progressive() {
        int16_t block[64];
        for (int c = 0; c < component_count; ++c) {
            Component& comp = components[c];
            const uint16_t* q = quantization[comp.tq];
            const int w = (comp.x + 7) >> 3, h = (comp.y + 7) >> 3;
            for (int by = 0; by < h; ++by)
                for (int bx = 0; bx < w; ++bx) {
                    const int16_t* coefficients = comp.coefficients.data() + 64 * (size_t(by) * comp.blocks_w + bx);
                    for (int i = 0; i < 64; ++i) block[i] = int16_t(coefficients[i] * q[i]);
                    inverse_dct(block, comp.plane.data() + size_t(by) * 8 * comp.plane_w + size_t(bx) * 8, comp.plane_w);
                }
        }
    }

    bool decode() {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file");
        pos = 2;
        bool scans = false;
        int pending = -1;
        for (;;) {
            int m = pending;
            pending = -1;
            if (m < 0) {
                if (pos >= size) break;
                if (read_u8() != 0xFF) continue;      // garbage between segments is skipped
                do m = read_u8(); while (m == 0xFF && pos < size);
                if (m == 0) continue;
            }
            if (m == 0xD9) break;                                   // EOI
            if (m >= 0xD0 && m <= 0xD7) continue;                   // stray restart marker
            const int length = read_u16();
            if (length < 2 || pos + size_t(length) - 2 > size) return fail("truncated segment");
            const size_t segment_end = pos + size_t(length) - 2;
            switch (m) {
            case 0xDB: if (!read_quantization_tables(length - 2)) return false; break;
            case 0xC4: if (!read_huffman_tables(length - 2)) return false; break;
            case 0xC0: case 0xC1: if (!read_frame(length, false)) return false; break;
            case 0xC2: if (!read_frame(length, true)) return false; break;
            case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF: return fail("lossless, hierarchical and arithmetic JPEG are not supported");
            case 0xDD: restart_interval = read_u16(); break;
            case 0xE0: jfif = length >= 7 && !std::memcmp(data + pos, "JFIF", 5); break;
            case 0xEE: if (length >= 14 && !std::memcmp(data + pos, "Adobe", 6)) adobe_transform = data[pos + 11]; break;
            case 0xDA:
                if (!frame_seen) return fail("scan before frame");
                if (!read_scan_header(length)) return false;
                pos = segment_end;
                if (!decode_scan()) return false;
                scans = true;
                if (hit_marker) pending = marker;
                else {      // the scan's blocks are done before its data is: find the marker that ends the segment
                    while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0 && !(data[pos + 1] >= 0xD0 && data[pos + 1] <= 0xD7))) ++pos;
                }
                continue;
            default: break;     // APPn, COM, DNL ...: skipped
            }
            pos = segment_end;
        }
        if (!frame_seen || !scans) return fail("no image data");
        if (progressive) finish_progressive();
        return error == nullptr;
    }

    // ---- upsampling + colour conversion into 1 or 3 channel rows, top-down -----------------------------------------------------------------
    static void upsample_row(const Component& comp, int hs, int vs, const uint8_t* near_row, const uint8_t* far_row, int lores_w, uint8_t* out) {
        if (hs == 1 && vs == 1) { std::memcpy(out, near_row, size_t(lores_w)); return; }
        if (hs == 1 && vs == 2) { for (int i = 0; i < lores_w; ++i) out[i] = uint8_t((3 * near_row[i] + far_row[i] + 2) >> 2); return; }
        if (hs == 2 && vs == 1) {
            const uint8_t* in = near_row;
            if (lores_w == 1) { out[0] = out[1] = in[0]; return; }
            out[0] = in[0];
            out[1] = uint8_t((in[0] * 3 + in[1] + 2) >> 2);
            for (int i = 1; i < lores_w - 1; ++i) {
                const int n = 3 * in[i] + 2;
                out[2 * i] = uint8_t((n + in[i - 1]) >> 2);
                out[2 * i + 1] = uint8_t((n + in[i + 1]) >> 2);
            }
            out[2 * (lores_w - 1)] = uint8_t((in[lores_w - 2] * 3 + in[lores_w - 1] + 2) >> 2);
            out[2 * (lores_w - 1) + 1] = in[lores_w - 1];
            return;
        }
        if (hs == 2 && vs == 2) {
            if (lores_w == 1) { out[0] = out[1] = uint8_t((3 * near_row[0] + far_row[0] + 2) >> 2); return; }
            int previous, current = 3 * near_row[0] + far_row[0];
            out[0] = uint8_t((current + 2) >> 2);
            for (int i = 1; i < lores_w; ++i) {
                previous = current;
                current = 3 * near_row[i] + far_row[i];
                out[2 * i - 1] = uint8_t((3 * previous + current + 8) >> 4);
                out[2 * i] = uint8_t((3 * current + previous + 8) >> 4);
            }
            out[2 * lores_w - 1] = uint8_t((current + 2) >> 2);
            return;
        }
        for (int i = 0; i < lores_w; ++i)       // other ratios: nearest
            for (int j = 0; j < hs; ++j) out[i * hs + j] = near_row[i];
        (void)comp;
    }

    void convert(std::vector<uint8_t>& pixels, unsigned& channels) const {
        channels = component_count >= 3 ? 3u : 1u;      // four components (CMYK / YCCK) come out as RGB
        pixels.assign(size_t(width) * height * channels, 0);
        const bool named_rgb = component_count == 3 && components[0].id == 'R' && components[1].id == 'G' && components[2].id == 'B';
        const bool is_rgb = component_count == 3 && (named_rgb || (adobe_transform == 0 && !jfif));      // component ids R, G, B, or an Adobe marker that says "no transform"
        struct RowState { int hs, vs, step, y, lores_w; size_t near_offset, far_offset; };      // far_offset: the row the interpolation leans towards
        RowState state[4];
        std::vector<uint8_t> line[4];
        for (int c = 0; c < component_count; ++c) {
            const Component& comp = components[c];
            state[c] = {h_max / comp.h, v_max / comp.v, (v_max / comp.v) >> 1, 0, (width + h_max / comp.h - 1) / (h_max / comp.h), 0, 0};
            line[c].assign(size_t(width) + 8, 0);
        }
        for (int j = 0; j < height; ++j) {
            for (int c = 0; c < component_count; ++c) {
                const Component& comp = components[c];
                RowState& s = state[c];
                // rows alternate between leaning on the previous and the next low-resolution row (the "line0 / line1" walk of the reference decoder)
                const bool lower_half = s.step >= (s.vs >> 1);
                const uint8_t* line0 = comp.plane.data() + s.near_offset;
                const uint8_t* line1 = comp.plane.data() + s.far_offset;
                upsample_row(comp, s.hs, s.vs, lower_half ? line1 : line0, lower_half ? line0 : line1, s.lores_w, line[c].data());
                if (++s.step >= s.vs) {
                    s.step = 0;
                    s.near_offset = s.far_offset;
                    if (++s.y < comp.y) s.far_offset += size_t(comp.plane_w);
                }
            }
            uint8_t* out = pixels.data() + size_t(j) * width * channels;
            if (channels == 1) { std::memcpy(out, line[0].data(), size_t(width)); continue; }
            if (is_rgb) {
                for (int i = 0; i < width; ++i) { out[3 * i] = line[0][i]; out[3 * i + 1] = line[1][i]; out[3 * i + 2] = line[2][i]; }
                continue;
            }
            // Four components (Adobe): transform 0 = CMYK stored inverted, every ink times black; 2 = YCCK, the YCbCr conversion below gives inverted
            // CMY which is then inverted and multiplied by black; anything else: YCbCr with a fourth channel that is ignored.
            auto times = [](unsigned x, unsigned y) { const unsigned t = x * y + 128u; return uint8_t((t + (t >> 8)) >> 8); };     // (x * y) / 255, rounded
            if (component_count == 4 && adobe_transform == 0) {
                for (int i = 0; i < width; ++i) { const unsigned k = line[3][i]; out[3 * i] = times(line[0][i], k); out[3 * i + 1] = times(line[1][i], k); out[3 * i + 2] = times(line[2][i], k); }
                continue;
            }
            for (int i = 0; i < width; ++i) {      // 20 bit fixed point; the Cb term of green is truncated to 16 bits as in the reference's SIMD-exact scalar kernel
                const int y_fixed = (line[0][i] << 20) + (1 << 19), cb = line[1][i] - 128, cr = line[2][i] - 128;
                const int kr = int(1.40200f * 4096.0f + 0.5f) << 8, kg_cr = int(0.71414f * 4096.0f + 0.5f) << 8, kg_cb = int(0.34414f * 4096.0f + 0.5f) << 8, kb = int(1.77200f * 4096.0f + 0.5f) << 8;
                const int r = (y_fixed + cr * kr) >> 20;
                const int g = (y_fixed + cr * -kg_cr + int(uint32_t(cb * -kg_cb) & 0xffff0000u)) >> 20;
                const int b = (y_fixed + cb * kb) >> 20;
                out[3 * i] = clamp_u8(r); out[3 * i + 1] = clamp_u8(g); out[3 * i + 2] = clamp_u8(b);
            }
            if (component_count == 4 && adobe_transform == 2)
                for (int i = 0; i < width; ++i) { const unsigned k = line[3][i]; out[3 * i] = times(255u - out[3 * i], k); out[3 * i + 1] = times(255u - out[3 * i + 1], k); out[3 * i + 2] = times(255u - out[3 * i + 2], k); }
        }
    }
};

} // namespace

bool is_jpeg(const void* data, size_t byte_count) {
    const uint8_t* b = static_cast<const uint8_t*>(data);
    return byte_count >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF;
}

bool decode(const void* data, size_t byte_count, unsigned& width, unsigned& height, unsigned& channels, std::vector<uint8_t>& pixels, std::string* error) {
    Decoder d;
    d.data = static_cast<const uint8_t*>(data);
    d.size = byte_count;
    std::memset(d.quantization, 0, sizeof(d.quantization));
    if (!d.decode()) {
        if (error) *error = d.error ? d.error : "corrupt JPEG";
        return false;
    }
    width = unsigned(d.width); height = unsigned(d.height);
    d.convert(pixels, channels);
    return true;
}

static Image to_image(const std::string& name, const void* data, size_t byte_count, bool flip_rows) {
    unsigned width = 0, height = 0, channels = 0;
    std::vector<uint8_t> pixels;
    std::string error;
    if (!decode(data, byte_count, width, height, channels, pixels, &error)) {
        printf("JpegImage::load(%s) error: '%s'\n", name.c_str(), error.c_str());
        return Image();
    }
    // StbImageLoader.cpp:26-47: 8 bit images are sRGB; 1 channel -> Intensity8, 3 -> RGB24
    Image image = Image::create2D(name, channels == 1 ? PixelFormat::Intensity8 : PixelFormat::RGB24, true, width, height);
    uint8_t* out = image.get_pixels<uint8_t>();
    const size_t row = size_t(width) * channels;
    for (unsigned y = 0; y < height; ++y) std::memcpy(out + size_t(y) * row, pixels.data() + size_t(flip_rows ? height - 1 - y : y) * row, row);
    return image;
}

Image load(const std::string& path) {
    std::ifstream file(path, std::ios::binary);
    if (!file) { printf("JpegImage::load(%s) error: 'could not read the file'\n", path.c_str()); return Image(); }
    std::vector<char> bytes((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    return to_image(path, bytes.data(), bytes.size(), true);
}

Image load_from_memory(const std::string& name, const void* data, size_t byte_count) { return to_image(name, data, byte_count, false); }

} // namespace JpegImage
