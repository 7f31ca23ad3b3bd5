// Host-side mirror of the objects the reference's game.cpp drives, on top of the C ABI (mrt.h).
//
//   reference (namespace Tmpl8)                        here (namespace mrt_host)
//   Surface   surface.h:57-100  ARGB framebuffer       Surface   (pixels, width, height, pitch)
//   Camera    scenedata.h:125-137                      Camera    (transform, SetPosition, LookAt, Get*)
//   Scene     scenedata.h:98-119  Add(file, scale)     Scene     (Add: .obj via LoadOBJ/LoadMTL/.bin, or .mrts)
//   Mesh / Texture  scenedata.h:39-52, 143-167         Mesh / Texture
//   Raytracer raytracer.h:217-245 Init/BuildBVH/Render Raytracer (+ RenderRecursive for GetColorFromSphere)
//   Rasterizer rasterizer.h:22-37 Init/Render          Rasterizer
//
// Everything that computes pixels happens behind mrt_* calls on the GPU; this header only holds
// scene data, parses files and forwards.  Header-only so that a driver needs nothing but libmrt.so.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#include <algorithm>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/mrt.h"

namespace mrt_host {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };

struct Surface {                       // surface.h:57-100 (pixel storage only)
    int width, height, pitch;
    uint32_t* buffer;                  // page-locked when a GPU is present (the frame is then written straight into it)
    bool pinned;
    Surface(int w, int h) : width(w), height(h), pitch(w) {
        const size_t bytes = (size_t)w * h * sizeof(uint32_t);
        buffer = static_cast<uint32_t*>(mrt_host_alloc(bytes));
        pinned = buffer != nullptr;
        if (!pinned) buffer = static_cast<uint32_t*>(aligned_alloc(64, (bytes + 63) / 64 * 64));   // surface.cpp:114-120
        if (!buffer) throw std::bad_alloc();
        Clear(0);
    }
    ~Surface() { if (pinned) mrt_host_free(buffer); else free(buffer); }
    Surface(const Surface&) = delete;
    Surface& operator=(const Surface&) = delete;
    uint32_t* GetBuffer() { return buffer; }
    int GetWidth() const { return width; }
    int GetHeight() const { return height; }
    int GetPitch() const { return pitch; }
    void Clear(uint32_t c) { std::fill(buffer, buffer + (size_t)pitch * height, c); }      // surface.cpp Clear
    bool SavePPM(const std::string& path) const {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        fprintf(f, "P6\n%d %d\n255\n", width, height);
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {
                const uint32_t p = buffer[(size_t)y * pitch + x];
                const unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p};
                fwrite(rgb, 1, 3, f);
            }
        fclose(f);
        return true;
    }
    // Surface::Print (surface.cpp:175-199): 5 x 5 pixel glyphs, 6 pixels apart, upper case drawn as lower case, only
    // the set dots of a glyph touch the frame.  The glyph shapes here are this file's own (digits, letters and the
    // punctuation the overlay and a driver's messages need; anything else prints as a blank), and, unlike the
    // reference, dots that would fall outside the frame are dropped instead of written.
    void Print(const char* text, int x1, int y1, uint32_t color) {
        for (int i = 0; text[i]; ++i) {
            const uint32_t g = Glyph(text[i]);
            for (int v = 0; v < 5; ++v)
                for (int h = 0; h < 5; ++h) {
                    const int x = x1 + 6 * i + h, y = y1 + v;
                    if (((g >> (24 - (5 * v + h))) & 1u) && x >= 0 && y >= 0 && x < width && y < height) buffer[(size_t)y * pitch + x] = color;
                }
        }
    }
    // The reference's per-frame overlay (game.cpp:95-112): two lines at (2,2) and (2,12) in 0xff00ff.
    void PrintFrameStats(float frame_duration_s, int frames_per_second) {
        char line[64];
        snprintf(line, sizeof line, "Frametime : %f", frame_duration_s);
        Print(line, 2, 2, 0xff00ff);
        snprintf(line, sizeof line, "FPS       : %i", frames_per_second);
        Print(line, 2, 12, 0xff00ff);
    }
    // The frame as a PNG file (8-bit RGB, alpha dropped like the PPM): what stands in for the window the reference
    // presents its Surface in (template.cpp:111-127).  No compression library: the image data is a zlib stream of
    // stored blocks.
    bool SavePNG(const std::string& path) const {
        std::vector<unsigned char> raw((size_t)height * (1 + 3 * (size_t)width));
        for (int y = 0; y < height; ++y) {
            unsigned char* row = raw.data() + (size_t)y * (1 + 3 * (size_t)width);
            row[0] = 0;                                        // filter type: none
            for (int x = 0; x < width; ++x) {
                const uint32_t p = buffer[(size_t)y * pitch + x];
                row[1 + 3 * x] = (unsigned char)(p >> 16); row[2 + 3 * x] = (unsigned char)(p >> 8); row[3 + 3 * x] = (unsigned char)p;
            }
        }
        std::vector<unsigned char> z = {0x78, 0x01};           // zlib header: deflate, 32 KiB window, no preset dictionary
        uint32_t a = 1, b = 0;                                 // Adler-32 of the uncompressed data
        for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
            const size_t n = std::min<size_t>(65535, raw.size() - off);
            const bool last = off + n >= raw.size();
            z.push_back(last ? 1 : 0);                         // stored block
            z.push_back((unsigned char)(n & 255)); z.push_back((unsigned char)(n >> 8));
            z.push_back((unsigned char)(~n & 255)); z.push_back((unsigned char)((~n >> 8) & 255));
            z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
            for (size_t i = off; i < off + n; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
            if (last) break;
        }
        const uint32_t adler = (b << 16) | a;
        for (int s = 24; s >= 0; s -= 8) z.push_back((unsigned char)(adler >> s));
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) return false;
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        bool ok = fwrite(sig, 1, 8, f) == 8;
        unsigned char ihdr[13];
        PutBE(ihdr, (uint32_t)width); PutBE(ihdr + 4, (uint32_t)height);
        ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;   // 8 bits, RGB, deflate, adaptive filters, no interlace
        ok = ok && Chunk(f, "IHDR", ihdr, 13) && Chunk(f, "IDAT", z.data(), z.size()) && Chunk(f, "IEND", nullptr, 0);
        return fclose(f) == 0 && ok;
    }

private:
    static void PutBE(unsigned char* p, uint32_t v) { p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v; }
    static uint32_t Crc(uint32_t crc, const unsigned char* p, size_t n) {      // CRC-32 (reflected 0xedb88320), bit by bit
        for (size_t i = 0; i < n; ++i) {
            crc ^= p[i];
            for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u)));
        }
        return crc;
    }
    static bool Chunk(FILE* f, const char type[4], const unsigned char* data, size_t n) {
        unsigned char len[4], tail[4];
        PutBE(len, (uint32_t)n);
        uint32_t crc = Crc(0xffffffffu, reinterpret_cast<const unsigned char*>(type), 4);
        if (n) crc = Crc(crc, data, n);
        PutBE(tail, ~crc);
        return fwrite(len, 1, 4, f) == 4 && fwrite(type, 1, 4, f) == 4 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tail, 1, 4, f) == 4;
    }
    // 25 bits per glyph, row by row from the top, the leftmost dot of a row in the higher bit
    static constexpr uint32_t Rows(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t r4) { return (r0 << 20) | (r1 << 15) | (r2 << 10) | (r3 << 5) | r4; }
    static uint32_t Glyph(char ch) {
        if (ch >= 'A' && ch <= 'Z') ch = (char)(ch - 'A' + 'a');
        switch (ch) {
            case '0': return Rows(0b01110, 0b10011, 0b10101, 0b11001, 0b01110);
            case '1': return Rows(0b00100, 0b01100, 0b00100, 0b00100, 0b01110);
            case '2': return Rows(0b11110, 0b00001, 0b01110, 0b10000, 0b11111);
            case '3': return Rows(0b11110, 0b00001, 0b01110, 0b00001, 0b11110);
            case '4': return Rows(0b10010, 0b10010, 0b11111, 0b00010, 0b00010);
            case '5': return Rows(0b11111, 0b10000, 0b11110, 0b00001, 0b11110);
            case '6': return Rows(0b01110, 0b10000, 0b11110, 0b10001, 0b01110);
            case '7': return Rows(0b11111, 0b00001, 0b00010, 0b00100, 0b00100);
            case '8': return Rows(0b01110, 0b10001, 0b01110, 0b10001, 0b01110);
            case '9': return Rows(0b01110, 0b10001, 0b01111, 0b00001, 0b01110);
            case 'a': return Rows(0b01110, 0b10001, 0b11111, 0b10001, 0b10001);
            case 'b': return Rows(0b11110, 0b10001, 0b11110, 0b10001, 0b11110);
            case 'c': return Rows(0b01111, 0b10000, 0b10000, 0b10000, 0b01111);
            case 'd': return Rows(0b11110, 0b10001, 0b10001, 0b10001, 0b11110);
            case 'e': return Rows(0b11111, 0b10000, 0b11110, 0b10000, 0b11111);
            case 'f': return Rows(0b11111, 0b10000, 0b11100, 0b10000, 0b10000);
            case 'g': return Rows(0b01111, 0b10000, 0b10011, 0b10001, 0b01110);
            case 'h': return Rows(0b10001, 0b10001, 0b11111, 0b10001, 0b10001);
            case 'i': return Rows(0b01110, 0b00100, 0b00100, 0b00100, 0b01110);
            case 'j': return Rows(0b00111, 0b00010, 0b00010, 0b10010, 0b01100);
            case 'k': return Rows(0b10010, 0b10100, 0b11000, 0b10100, 0b10010);
            case 'l': return Rows(0b10000, 0b10000, 0b10000, 0b10000, 0b11111);
            case 'm': return Rows(0b10001, 0b11011, 0b10101, 0b10001, 0b10001);
            case 'n': return Rows(0b10001, 0b11001, 0b10101, 0b10011, 0b10001);
            case 'o': return Rows(0b01110, 0b10001, 0b10001, 0b10001, 0b01110);
            case 'p': return Rows(0b11110, 0b10001, 0b11110, 0b10000, 0b10000);
            case 'q': return Rows(0b01110, 0b10001, 0b10001, 0b10011, 0b01111);
            case 'r': return Rows(0b11110, 0b10001, 0b11110, 0b10100, 0b10010);
            case 's': return Rows(0b01111, 0b10000, 0b01110, 0b00001, 0b11110);
            case 't': return Rows(0b11111, 0b00100, 0b00100, 0b00100, 0b00100);
            case 'u': return Rows(0b10001, 0b10001, 0b10001, 0b10001, 0b01110);
            case 'v': return Rows(0b10001, 0b10001, 0b01010, 0b01010, 0b00100);
            case 'w': return Rows(0b10001, 0b10001, 0b10101, 0b11011, 0b10001);
            case 'x': return Rows(0b10001, 0b01010, 0b00100, 0b01010, 0b10001);
            case 'y': return Rows(0b10001, 0b01010, 0b00100, 0b00100, 0b00100);
            case 'z': return Rows(0b11111, 0b00010, 0b00100, 0b01000, 0b11111);
            case ':': return Rows(0b00000, 0b00100, 0b00000, 0b00100, 0b00000);
            case '.': return Rows(0b00000, 0b00000, 0b00000, 0b00000, 0b00100);
            case ',': return Rows(0b00000, 0b00000, 0b00000, 0b00100, 0b01000);
            case '-': return Rows(0b00000, 0b00000, 0b01110, 0b00000, 0b00000);
            case '+': return Rows(0b00000, 0b00100, 0b01110, 0b00100, 0b00000);
            case '/': return Rows(0b00001, 0b00010, 0b00100, 0b01000, 0b10000);
            case '%': return Rows(0b11001, 0b11010, 0b00100, 0b01011, 0b10011);
            case '(': return Rows(0b00010, 0b00100, 0b00100, 0b00100, 0b00010);
            case ')': return Rows(0b01000, 0b00100, 0b00100, 0b00100, 0b01000);
            default: return 0;
        }
    }

public:
};

struct Camera {                        // scenedata.h:125-137; transform is column-major like glm::mat4
    float transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    void SetPosition(vec3 p) { transform[12] = p.x; transform[13] = p.y; transform[14] = p.z; }
    vec3 GetPosition() const { return {transform[12], transform[13], transform[14]}; }
    vec3 GetRight() const { return {transform[0], transform[1], transform[2]}; }
    vec3 GetUp() const { return {transform[4], transform[5], transform[6]}; }
    vec3 GetForward() const { return {-transform[8], -transform[9], -transform[10]}; }
    void LookAt(vec3 target) {
        const float p[3] = {transform[12], transform[13], transform[14]}, t[3] = {target.x, target.y, target.z};
        mrt_camera_look_at(p, t, transform);
    }
};

struct Texture {                       // scenedata.h:39-52 + surface.h:41-55
    std::string name;
    int width = 0, height = 0;
    std::vector<uint8_t> indices;
    std::vector<uint32_t> palettes;    // 32 x 256
};

struct Mesh {                          // scenedata.h:143-167 (arrays the tracer and the rasterizer read)
    std::vector<vec3> pos, norm;
    std::vector<vec2> uv;
    std::vector<vec3> N;
    std::vector<int32_t> tri;
    int texture = -1;
    float refl = 0.0f, refr = 0.0f, rindex = 1.0f;
    int verts() const { return (int)pos.size(); }
    int tris() const { return (int)N.size(); }
};

class Scene {                          // scenedata.h:98-119
public:
    std::vector<Mesh> meshList;
    std::vector<Texture> texList;

    // Scene::Add(file, scale): an .obj goes through LoadOBJ (+ LoadMTL and the textures' .bin
    // caches); a .mrts container is read as is.  Throws std::runtime_error on missing files
    // (the reference pops a message box and exits: template.cpp:18-23).
    void Add(const std::string& file, float scale = 1.0f) {
        if (file.size() > 5 && file.substr(file.size() - 5) == ".mrts") LoadMRTS(file);
        else LoadOBJ(file, scale);
    }

    // raytracer.cpp:908-929: the first mesh is the refracting maze, every later mesh is opaque
    void ApplyReferenceMaterials() {
        for (size_t i = 0; i < meshList.size(); ++i) {
            Mesh& m = meshList[i];
            m.refl = 0.0f; m.refr = i == 0 ? 0.3f : 0.0f; m.rindex = i == 0 ? 1.5f : 1.0f;
        }
    }

    void SaveMRTS(const std::string& path) const {
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        const uint32_t head[3] = {1u, (uint32_t)texList.size(), (uint32_t)meshList.size()};
        fwrite("MRTS", 1, 4, f); fwrite(head, 4, 3, f);
        for (const Texture& t : texList) {
            const int32_t wh[2] = {t.width, t.height};
            fwrite(wh, 4, 2, f); fwrite(t.indices.data(), 1, t.indices.size(), f); fwrite(t.palettes.data(), 4, 32 * 256, f);
        }
        for (const Mesh& m : meshList) {
            const int32_t h3[3] = {m.verts(), m.tris(), m.texture};
            const float mat[3] = {m.refl, m.refr, m.rindex};
            float b[6] = {1e30f, 1e30f, 1e30f, -1e30f, -1e30f, -1e30f};
            for (const vec3& p : m.pos) {
                b[0] = p.x < b[0] ? p.x : b[0]; b[1] = p.y < b[1] ? p.y : b[1]; b[2] = p.z < b[2] ? p.z : b[2];
                b[3] = p.x > b[3] ? p.x : b[3]; b[4] = p.y > b[4] ? p.y : b[4]; b[5] = p.z > b[5] ? p.z : b[5];
            }
            fwrite(h3, 4, 3, f); fwrite(mat, 4, 3, f); fwrite(b, 4, 6, f);
            fwrite(m.pos.data(), 12, m.pos.size(), f); fwrite(m.uv.data(), 8, m.uv.size(), f);
            fwrite(m.norm.data(), 12, m.norm.size(), f); fwrite(m.N.data(), 12, m.N.size(), f);
            fwrite(m.tri.data(), 4, m.tri.size(), f);
        }
        fclose(f);
    }

private:
    static bool ReadAll(FILE* f, void* dst, size_t bytes) { return bytes == 0 || fread(dst, bytes, 1, f) == 1; }

    void LoadMRTS(const std::string& path) {
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("File not found: " + path);
        struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};     // also when a resize below throws
        char magic[4];
        uint32_t head[3];
        bool ok = ReadAll(f, magic, 4) && !memcmp(magic, "MRTS", 4) && ReadAll(f, head, 12) && head[0] == 1;
        const int tex_base = (int)texList.size();
        for (uint32_t i = 0; ok && i < head[1]; ++i) {
            Texture t;
            int32_t wh[2];
            ok = ReadAll(f, wh, 8) && wh[0] > 0 && wh[1] > 0 && (int64_t)wh[0] * wh[1] < (1ll << 30);
            if (!ok) break;
            t.width = wh[0]; t.height = wh[1];
            t.indices.resize((size_t)wh[0] * wh[1]); t.palettes.resize(32 * 256);
            ok = ReadAll(f, t.indices.data(), t.indices.size()) && ReadAll(f, t.palettes.data(), 4 * 32 * 256);
            texList.push_back(std::move(t));
        }
        for (uint32_t i = 0; ok && i < head[2]; ++i) {
            Mesh m;
            int32_t h3[3];
            float mat[3], b[6];
            ok = ReadAll(f, h3, 12) && ReadAll(f, mat, 12) && ReadAll(f, b, 24) && h3[0] >= 0 && h3[1] >= 0 &&
                 h3[0] < (1 << 30) && h3[1] < (1 << 30);
            if (!ok) break;
            m.texture = h3[2] < 0 ? -1 : h3[2] + tex_base;
            m.refl = mat[0]; m.refr = mat[1]; m.rindex = mat[2];
            m.pos.resize(h3[0]); m.uv.resize(h3[0]); m.norm.resize(h3[0]); m.N.resize(h3[1]); m.tri.resize(3 * (size_t)h3[1]);
            ok = ReadAll(f, m.pos.data(), 12 * m.pos.size()) && ReadAll(f, m.uv.data(), 8 * m.uv.size()) &&
                 ReadAll(f, m.norm.data(), 12 * m.norm.size()) && ReadAll(f, m.N.data(), 12 * m.N.size()) &&
                 ReadAll(f, m.tri.data(), 4 * m.tri.size());
            meshList.push_back(std::move(m));
        }
        if (!ok) throw std::runtime_error("malformed scene container: " + path);
    }

    // surface.cpp:39-59: "<image minus extension>.bin" = int32 w, h, w*h index bytes, 32 x 256 uint32
    int FindOrLoadTexture(const std::string& image) {
        for (size_t i = 0; i < texList.size(); ++i) if (texList[i].name == image) return (int)i;
        const size_t dot = image.find_last_of('.');
        const std::string bin = (dot == std::string::npos || dot == 0 ? image : image.substr(0, dot)) + ".bin";
        FILE* f = fopen(bin.c_str(), "rb");
        if (!f) throw std::runtime_error("File not found: " + bin + " (only pre-quantised .bin texture caches are read)");
        Texture t;
        t.name = image;
        int32_t wh[2] = {0, 0};
        bool ok = ReadAll(f, wh, 8) && wh[0] > 0 && wh[1] > 0 && (int64_t)wh[0] * wh[1] < (1ll << 30);
        if (ok) {
            t.width = wh[0]; t.height = wh[1];
            t.indices.resize((size_t)wh[0] * wh[1]); t.palettes.resize(32 * 256);
            ok = ReadAll(f, t.indices.data(), t.indices.size()) && ReadAll(f, t.palettes.data(), 4 * 32 * 256);
        }
        fclose(f);
        if (!ok) throw std::runtime_error("malformed texture cache: " + bin);
        texList.push_back(std::move(t));
        return (int)texList.size() - 1;
    }

    // scenedata.cpp:77-111: "newmtl <name>" opens a material, "map_Kd <file>" names its texture,
    // looked up under <scene dir>/textures/.  A missing .mtl is silently ignored there as here.
    void LoadMTL(const std::string& dir, const std::string& file, std::map<std::string, int>& material_texture) {
        FILE* f = fopen((dir + file).c_str(), "r");
        if (!f) return;
        char line[1024];
        std::string current;
        while (fgets(line, sizeof line, f)) {
            size_t len = strlen(line);
            while (len && (unsigned char)line[len - 1] < 32) line[--len] = 0;
            char cmd[256] = "", arg[512] = "";
            if (sscanf(line, "%255s", cmd) != 1) continue;
            if (!strcasecmp(cmd, "newmtl")) {
                if (sscanf(line + strlen(cmd), "%511s", arg) == 1) { current = arg; material_texture[current] = -1; }
            } else if (!strcasecmp(cmd, "map_Kd") && !current.empty()) {
                const char* sp = strchr(line, ' ');
                if (sp && sp[1]) material_texture[current] = FindOrLoadTexture(dir + "textures/" + (sp + 1));
            }
        }
        fclose(f);
    }

    // scenedata.cpp:121-271.  What the tracer depends on, and what is reproduced exactly:
    //  - a mesh starts at every "usemtl" and takes the faces up to the next one;
    //  - vertices of a mesh are the distinct (position, normal, uv) index triples of its faces in
    //    order of first use; uv.v is flipped (1 - v); positions are scaled;
    //  - the face normal is normalize(cross(v1-v0, v2-v0)), negated when it opposes the vertex
    //    normal of the face's SECOND corner.
    void LoadOBJ(const std::string& file, float scale) {
        FILE* f = fopen(file.c_str(), "r");
        if (!f) throw std::runtime_error("File not found: " + file);
        const size_t slash = file.find_last_of("/\\");
        const std::string dir = slash == std::string::npos ? std::string() : file.substr(0, slash + 1);
        std::map<std::string, int> material_texture;
        std::vector<vec3> v, vn;
        std::vector<vec2> vt;
        struct Corner { int v, t, n; };
        std::vector<Corner> corners;            // faces of the mesh being collected
        bool collecting = false, has_uv = false;
        int texture = -1;
        auto flush = [&]() {
            if (!collecting) return;
            Mesh m;
            m.texture = texture;
            std::map<std::tuple<int, int, int>, int> seen;
            for (const Corner& c : corners) {
                auto key = std::make_tuple(c.v, c.n, c.t);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    it = seen.emplace(key, (int)m.pos.size()).first;
                    m.pos.push_back(c.v >= 0 && c.v < (int)v.size() ? v[c.v] : vec3{0, 0, 0});
                    m.norm.push_back(c.n >= 0 && c.n < (int)vn.size() ? vn[c.n] : vec3{0, 0, 0});
                    m.uv.push_back(has_uv && c.t >= 0 && c.t < (int)vt.size() ? vt[c.t] : vec2{0, 1});
                }
                m.tri.push_back(it->second);
            }
            const int nt = (int)m.tri.size() / 3;
            m.tri.resize(3 * (size_t)nt);
            m.N.resize(nt);
            for (int i = 0; i < nt; ++i) {
                const vec3 a = m.pos[m.tri[3 * i]], b = m.pos[m.tri[3 * i + 1]], c = m.pos[m.tri[3 * i + 2]];
                const vec3 e1 = {b.x - a.x, b.y - a.y, b.z - a.z}, e2 = {c.x - a.x, c.y - a.y, c.z - a.z};
                const vec3 cr = {e1.y * e2.z - e2.y * e1.z, e1.z * e2.x - e2.z * e1.x, e1.x * e2.y - e2.x * e1.y};
                const float sq = cr.x * cr.x + cr.y * cr.y + cr.z * cr.z;
                const float r = 1.0f / sqrtf_(sq);
                vec3 n = {cr.x * r, cr.y * r, cr.z * r};
                const vec3 vn1 = m.norm[m.tri[3 * i + 1]];
                const float px = n.x * vn1.x, py = n.y * vn1.y, pz = n.z * vn1.z;
                if (px + py + pz < 0) n = {n.x * -1.0f, n.y * -1.0f, n.z * -1.0f};
                m.N[i] = n;
            }
            meshList.push_back(std::move(m));
            corners.clear();
        };
        char line[2048];
        while (fgets(line, 1023, f)) {
            if (!strncasecmp(line, "mtllib", 6)) {
                char name[1024] = "";
                if (sscanf(line + 7, "%1023s", name) == 1) LoadMTL(dir, name, material_texture);
                has_uv = false;
            } else if (!strncasecmp(line, "usemtl", 6)) {
                flush();
                char name[1024] = "";
                sscanf(line + 7, "%1023s", name);
                auto it = material_texture.find(name);
                texture = it == material_texture.end() ? -1 : it->second;
                collecting = true;
                has_uv = false;
            } else if (line[0] == 'v' && line[1] == ' ') {
                vec3 p = {0, 0, 0};
                sscanf(line + 2, "%f %f %f", &p.x, &p.y, &p.z);
                v.push_back({p.x * scale, p.y * scale, p.z * scale});
            } else if (line[0] == 'v' && line[1] == 't') {
                vec2 t = {0, 0};
                sscanf(line + 3, "%f %f", &t.x, &t.y);
                t.y = 1.0f - t.y;
                vt.push_back(t);
            } else if (line[0] == 'v' && line[1] == 'n') {
                vec3 n = {0, 0, 0};
                sscanf(line + 3, "%f %f %f", &n.x, &n.y, &n.z);
                vn.push_back(n);
            } else if (line[0] == 'f' && collecting) {
                int a[3] = {0, 0, 0}, t[3] = {0, 0, 0}, n[3] = {0, 0, 0};
                if (strstr(line, "//")) sscanf(line + 2, "%i//%i %i//%i %i//%i", &a[0], &n[0], &a[1], &n[1], &a[2], &n[2]);
                else sscanf(line + 2, "%i/%i/%i %i/%i/%i %i/%i/%i", &a[0], &t[0], &n[0], &a[1], &t[1], &n[1], &a[2], &t[2], &n[2]);
                for (int i = 0; i < 3; ++i) {
                    if (t[i] > 0) has_uv = true;
                    corners.push_back({a[i] - 1, t[i] - 1, n[i] - 1});
                }
            }
        }
        flush();
        fclose(f);
    }
    static float sqrtf_(float x) { return __builtin_sqrtf(x); }
};

class Raytracer {                      // raytracer.h:217-245
public:
    Raytracer() = default;
    ~Raytracer() { mrt_destroy(ctx_); }
    Raytracer(const Raytracer&) = delete;
    Raytracer& operator=(const Raytracer&) = delete;

    // Raytracer::Init(screen) (raytracer.cpp:384-413); device -1 = current GPU
    void Init(Surface* screen, int device = -1) {
        screen_ = screen;
        check(mrt_create(screen->GetWidth(), screen->GetHeight(), device, &ctx_), "mrt_create");
    }
    // Raytracer::BuildBVH(meshList) (raytracer.cpp:670-673).  second_tree: upload with mrt_scene_upload_lbvh_exact -- the
    // reference's tree as ever, and beside it the GPU-built one that frames of the recursive shader walk with identical
    // results (mrt.h); false: mrt_scene_upload, the reference's tree alone.
    void BuildBVH(const Scene& scene, bool second_tree = true) {
        std::vector<mrt_mesh> meshes;
        for (const Mesh& m : scene.meshList) {
            if (m.tris() == 0) continue;       // a "usemtl" that no face follows leaves an empty mesh: nothing to trace
            meshes.push_back(mrt_mesh{reinterpret_cast<const float*>(m.pos.data()), reinterpret_cast<const float*>(m.uv.data()),
                                      reinterpret_cast<const float*>(m.N.data()), m.tri.data(), m.verts(), m.tris(), m.texture,
                                      m.refl, m.refr, m.rindex});
        }
        std::vector<mrt_texture> textures;
        for (const Texture& t : scene.texList) textures.push_back(mrt_texture{t.indices.data(), t.palettes.data(), t.width, t.height});
        if (second_tree) check(mrt_scene_upload_lbvh_exact(ctx_, meshes.data(), (int)meshes.size(), textures.data(), (int)textures.size()), "mrt_scene_upload_lbvh_exact");
        else check(mrt_scene_upload(ctx_, meshes.data(), (int)meshes.size(), textures.data(), (int)textures.size()), "mrt_scene_upload");
    }
    // m_Lights (raytracer.h:244): 7 floats per light {pos, colour in A,R,G,B order}
    void SetLights(const std::vector<float>& pos_argb7) {
        check(mrt_set_lights(ctx_, pos_argb7.data(), (int)(pos_argb7.size() / 7)), "mrt_set_lights");
    }
    // Raytracer::Render(camera) (raytracer.cpp:420-463): packets + direct light
    void Render(const Camera& camera) { render(camera, MRT_MODE_PACKET); }
    // the per-pixel GetColorFromSphere loop the reference keeps commented out (raytracer.cpp:465-490)
    void RenderRecursive(const Camera& camera) { render(camera, MRT_MODE_RECURSIVE); }
    // the BVH-depth view ('T' in game.cpp:73 toggles Raytracer::traverseDepth; raytracer.cpp:581-586)
    void RenderDepth(const Camera& camera) { render(camera, MRT_MODE_BVH_DEPTH); }
    // m_Objects (raytracer.h:243): spheres and planes traced next to the BVH
    void SetObjects(const std::vector<mrt_object>& objects) {
        check(mrt_set_objects(ctx_, objects.data(), (int)objects.size()), "mrt_set_objects");
    }
    // Rasterizer::Render(camera) shares the context (the reference's rasterizer owns the scene)
    void RenderRaster(const Camera& camera) {
        check(mrt_raster_render(ctx_, camera.transform, screen_->GetBuffer(), screen_->GetPitch(), nullptr, nullptr), "mrt_raster_render");
    }
    // The frame loop with the next frame in flight (template.cpp:111-127 presents the Surface after Tick):
    // Submit queues a frame into `target` (page-locked, as every Surface here is) and returns a ticket;
    // Wait returns when that frame has landed.
    int Submit(const Camera& camera, Surface* target, bool recursive) {
        int ticket = -1;
        check(mrt_render_submit(ctx_, camera.transform, recursive ? MRT_MODE_RECURSIVE : MRT_MODE_PACKET, target->GetBuffer(),
                                target->GetPitch(), &ticket), "mrt_render_submit");
        return ticket;
    }
    void Wait(int ticket) { check(mrt_render_wait(ctx_, ticket), "mrt_render_wait"); }
    void RayCounts(uint64_t out[4]) { check(mrt_get_ray_counts(ctx_, out), "mrt_get_ray_counts"); }
    void TileSchedule(uint64_t out[2]) { check(mrt_get_tile_schedule(ctx_, out), "mrt_get_tile_schedule"); }
    mrt_ctx* Context() { return ctx_; }

private:
    void render(const Camera& camera, uint32_t flags) {
        check(mrt_render(ctx_, camera.transform, flags, screen_->GetBuffer(), screen_->GetPitch(), nullptr), "mrt_render");
    }
    void check(int rc, const char* what) {
        if (rc != MRT_OK) throw std::runtime_error(std::string(what) + ": " + mrt_last_error(ctx_));
    }
    mrt_ctx* ctx_ = nullptr;
    Surface* screen_ = nullptr;
};

}  // namespace mrt_host
