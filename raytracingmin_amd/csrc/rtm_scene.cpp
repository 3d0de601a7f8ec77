// rtm_scene.cpp — host-side scene input: the reference's JSON schema read into the flat C structs
// of include/rtm.h.  Mirrors png::LoadData (reference src/SettingData.cpp:6-12, 14-24, 100-186).
//
// Own minimal JSON reader (the reference uses nlohmann/json, an empty submodule in the checkout).
// Loader rules so the shipped ExampleScene/*.json run unchanged:
//   missing "00 objectType" => 1 (sphere); objects without "00 position" are skipped (the "{}" in
//   cornellBoxSetting.json:47); "00 sample" is accepted for "00 samples"; missing samples /
//   superSamples default to 10 / 1; unknown keys are ignored like from_json does; objectType 2 is the
//   build-defined plane (rtm.h: rtm_object; keys "03 up", "04 target", "01 size" = width); any other objectType
//   is an error (HEAD would push a nullptr and crash at src/Renderer.cpp:66).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtm_internal.h"

namespace rtm {
namespace {

struct JsonError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct JsonValue {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    long long i = 0;
    double d = 0.0;
    std::string s;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;  // file order kept

    const JsonValue* find(const std::string& key) const {
        const JsonValue* hit = nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) hit = &kv.second;  // last duplicate wins, like nlohmann
        return hit;
    }
    bool is_number() const { return kind == Int || kind == Float; }
    double as_double(const char* what) const {
        if (kind == Int) return (double)i;
        if (kind == Float) return d;
        throw JsonError(std::string("type error: ") + what + " must be a number");
    }
    // nlohmann's get<int>() on a float value is a static_cast (truncation)
    int as_int(const char* what) const {
        if (kind == Int) return (int)i;
        if (kind == Float) return (int)d;
        throw JsonError(std::string("type error: ") + what + " must be a number");
    }
};

class JsonParser {
  public:
    JsonParser(const char* p, size_t n) : p_(p), end_(p + n) {}
    JsonValue parse() {
        JsonValue v = value(0);
        ws();
        if (p_ != end_) fail("trailing characters after the document");
        return v;
    }

  private:
    const char* p_;
    const char* end_;
    [[noreturn]] void fail(const std::string& m) const { throw JsonError("parse error: " + m); }
    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) ++p_;
    }
    bool lit(const char* s) {
        const size_t n = std::strlen(s);
        if ((size_t)(end_ - p_) >= n && std::memcmp(p_, s, n) == 0) {
            p_ += n;
            return true;
        }
        return false;
    }
    JsonValue value(int depth) {
        if (depth > 256) fail("nesting too deep");
        ws();
        if (p_ == end_) fail("unexpected end of input");
        JsonValue v;
        const char c = *p_;
        if (c == '{') {
            ++p_;
            v.kind = JsonValue::Object;
            ws();
            if (p_ < end_ && *p_ == '}') {
                ++p_;
                return v;
            }
            for (;;) {
                ws();
                if (p_ == end_ || *p_ != '"') fail("object key must be a string");
                std::string key = string();
                ws();
                if (p_ == end_ || *p_ != ':') fail("expected ':'");
                ++p_;
                v.obj.emplace_back(std::move(key), value(depth + 1));
                ws();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == '}') {
                    ++p_;
                    return v;
                }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p_;
            v.kind = JsonValue::Array;
            ws();
            if (p_ < end_ && *p_ == ']') {
                ++p_;
                return v;
            }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p_ < end_ && *p_ == ',') {
                    ++p_;
                    continue;
                }
                if (p_ < end_ && *p_ == ']') {
                    ++p_;
                    return v;
                }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = JsonValue::String;
            v.s = string();
            return v;
        }
        if (lit("true")) {
            v.kind = JsonValue::Bool;
            v.b = true;
            return v;
        }
        if (lit("false")) {
            v.kind = JsonValue::Bool;
            return v;
        }
        if (lit("null")) return v;
        if (c == '-' || (c >= '0' && c <= '9')) return number();
        fail(std::string("unexpected character '") + c + "'");
    }
    static void append_utf8(std::string& out, unsigned cp) {
        if (cp < 0x80)
            out += (char)cp;
        else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
    unsigned hex4() {
        if (end_ - p_ < 4) fail("short \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) {
            const char h = *p_++;
            v <<= 4;
            if (h >= '0' && h <= '9') v |= (unsigned)(h - '0');
            else if (h >= 'a' && h <= 'f') v |= (unsigned)(h - 'a' + 10);
            else if (h >= 'A' && h <= 'F') v |= (unsigned)(h - 'A' + 10);
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        ++p_;  // opening quote
        std::string out;
        for (;;) {
            if (p_ == end_) fail("unterminated string");
            const unsigned char c = (unsigned char)*p_++;
            if (c == '"') return out;
            if (c < 0x20) fail("control character in string");
            if (c != '\\') {
                out += (char)c;
                continue;
            }
            if (p_ == end_) fail("unterminated escape");
            const char e = *p_++;
            switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                        p_ += 2;
                        const unsigned lo = hex4();
                        if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else fail("bad surrogate pair");
                    }
                    append_utf8(out, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
    }
    JsonValue number() {
        const char* s = p_;
        bool is_float = false;
        if (*p_ == '-') ++p_;
        if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("bad number");
        if (*p_ == '0') ++p_;
        else while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        if (p_ < end_ && *p_ == '.') {
            is_float = true;
            ++p_;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("bad fraction");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            is_float = true;
            ++p_;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') fail("bad exponent");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
        }
        const std::string tok(s, p_);
        JsonValue v;
        if (!is_float) {
            errno = 0;
            char* e = nullptr;
            const long long iv = std::strtoll(tok.c_str(), &e, 10);
            if (errno == 0 && e && *e == 0) {
                v.kind = JsonValue::Int;
                v.i = iv;
                return v;
            }
        }
        v.kind = JsonValue::Float;
        v.d = std::strtod(tok.c_str(), nullptr);  // correctly rounded, like nlohmann's strtod path
        return v;
    }
};

void read_vec3(const JsonValue& v, const char* what, double out[3]) {
    if (v.kind != JsonValue::Array || v.arr.size() < 3)
        throw JsonError(std::string("type error: ") + what + " must be an array of 3 numbers");
    for (int k = 0; k < 3; ++k) out[k] = v.arr[k].as_double(what);
}

}  // namespace

// The parser proper: every object becomes an rtm_object (type 1 = sphere; type 2 = plane, a build-defined
// extension of the schema — the reference's loader builds only type 1 and pushes a null Object* for anything
// else, src/SettingData.cpp:161-179).
static int parse_json_objects(const char* text, size_t len, int literal_loader, rtm_settings* st,
                              std::vector<rtm_object>& objs) {
    try {
        const JsonValue root = JsonParser(text, len).parse();
        if (root.kind != JsonValue::Object) throw JsonError("type error: document must be an object");
        std::memset(st, 0, sizeof *st);
        st->samples = 10;       // HEAD leaves these indeterminate when the key is absent
        st->super_samples = 1;  // (src/SettingData.h:47-51); build defaults, SURVEY Appendix C
        bool have_w = false, have_h = false;
        // reference: for (auto& it : json.items()) key-by-key (src/SettingData.cpp:130-184)
        for (const auto& kv : root.obj) {
            const std::string& key = kv.first;
            const JsonValue& val = kv.second;
            if (key == "00 width") {
                st->width = val.as_int("00 width");
                have_w = true;
            } else if (key == "00 height") {
                st->height = val.as_int("00 height");
                have_h = true;
            } else if (key == "00 samples" || key == "00 sample") {
                st->samples = val.as_int("00 samples");
            } else if (key == "00 superSamples") {
                st->super_samples = val.as_int("00 superSamples");
            } else if (key == "01 camera") {
                if (val.kind != JsonValue::Object) throw JsonError("type error: 01 camera must be an object");
                for (const auto& ck : val.obj) {
                    if (ck.first == "origin") read_vec3(ck.second, "origin", st->camera.origin);
                    else if (ck.first == "target") read_vec3(ck.second, "target", st->camera.target);
                    else if (ck.first == "upVec") read_vec3(ck.second, "upVec", st->camera.up);
                    else if (ck.first == "fov") st->camera.fov = (float)ck.second.as_double("fov");
                }
            } else if (key == "02 scene") {
                if (val.kind != JsonValue::Object) throw JsonError("type error: 02 scene must be an object");
                for (const auto& sk : val.obj) {
                    if (sk.first != "00 object") continue;
                    if (sk.second.kind != JsonValue::Array && sk.second.kind != JsonValue::Object)
                        throw JsonError("type error: 00 object must be an array");
                    std::vector<const JsonValue*> items;
                    if (sk.second.kind == JsonValue::Array)
                        for (const auto& o : sk.second.arr) items.push_back(&o);
                    else
                        for (const auto& o : sk.second.obj) items.push_back(&o.second);
                    for (const JsonValue* o : items) {
                        if (o->kind != JsonValue::Object) throw JsonError("type error: scene object must be an object");
                        const JsonValue* pos = o->find("00 position");
                        if (!pos) continue;  // e.g. the "{}" entry of cornellBoxSetting.json
                        int object_type = 1;
                        if (const JsonValue* t = o->find("00 objectType")) object_type = t->as_int("00 objectType");
                        if (object_type != RTM_OBJECT_SPHERE && object_type != RTM_OBJECT_PLANE) {
                            set_last_error("objectType " + std::to_string(object_type) +
                                           " is not supported (1 = sphere, 2 = plane; the reference would "
                                           "dereference a null Object*)");
                            return RTM_ERR_INVALID_SCENE;
                        }
                        rtm_object s;
                        std::memset(&s, 0, sizeof s);
                        s.type = object_type;
                        double p[3];
                        read_vec3(*pos, "00 position", p);
                        if (literal_loader) {  // src/SettingData.cpp:165-167: posi.x = [0]; = [1]; = [2]
                            s.position[0] = p[2];
                            s.position[1] = 0.0;
                            s.position[2] = 0.0;
                        } else {
                            s.position[0] = p[0];
                            s.position[1] = p[1];
                            s.position[2] = p[2];
                        }
                        const JsonValue* size = o->find("01 size");
                        if (!size) throw JsonError("type error: object without \"01 size\"");
                        if (object_type == RTM_OBJECT_SPHERE) {
                            s.size = (float)size->as_double("01 size");  // double size -> const float size
                        } else {  // PlaneObject(position, up, target, const double width, mat)
                            s.width = size->as_double("01 size");
                            const JsonValue* up = o->find("03 up");
                            const JsonValue* tg = o->find("04 target");
                            if (!up || !tg) throw JsonError("type error: a plane needs \"03 up\" and \"04 target\"");
                            read_vec3(*up, "03 up", s.up);
                            read_vec3(*tg, "04 target", s.target);
                        }
                        const JsonValue* mat = o->find("02 material");
                        if (!mat || mat->kind != JsonValue::Object)
                            throw JsonError("type error: object without \"02 material\"");
                        const JsonValue* col = mat->find("color");
                        const JsonValue* emi = mat->find("emission");
                        if (!col || !emi) throw JsonError("type error: material needs color and emission");
                        read_vec3(*col, "color", s.color);
                        read_vec3(*emi, "emission", s.emission);
                        objs.push_back(s);
                    }
                }
            }
        }
        if (!have_w || !have_h) {
            set_last_error("\"00 width\" and \"00 height\" are required");
            return RTM_ERR_INVALID_SCENE;
        }
        return RTM_OK;
    } catch (const JsonError& e) {
        set_last_error(e.what());
        return RTM_ERR_PARSE;
    }
}

int scene_parse_json_objects(const char* text, size_t len, int literal_loader, rtm_settings* st,
                             rtm_object* objects, size_t capacity, size_t* n_objects) {
    if (!text || !st || !n_objects || (!objects && capacity)) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    std::vector<rtm_object> objs;
    const int rc = parse_json_objects(text, len, literal_loader, st, objs);
    if (rc != RTM_OK) return rc;
    *n_objects = objs.size();
    if (objects) {
        if (objs.size() > capacity) {
            set_last_error("object buffer too small");
            return RTM_ERR_CAPACITY;
        }
        if (!objs.empty()) std::memcpy(objects, objs.data(), objs.size() * sizeof(rtm_object));
    }
    return RTM_OK;
}

int scene_parse_json(const char* text, size_t len, int literal_loader, rtm_settings* st,
                     rtm_sphere* spheres, size_t capacity, size_t* n_spheres) {
    if (!text || !st || !n_spheres || (!spheres && capacity)) {
        set_last_error("null argument");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    std::vector<rtm_object> objs;
    const int rc = parse_json_objects(text, len, literal_loader, st, objs);
    if (rc != RTM_OK) return rc;
    for (const rtm_object& o : objs)
        if (o.type != RTM_OBJECT_SPHERE) {
            set_last_error("the scene holds a plane (objectType 2): load it with rtm_scene_load_json_objects");
            return RTM_ERR_INVALID_SCENE;
        }
    *n_spheres = objs.size();
    if (spheres) {
        if (objs.size() > capacity) {
            set_last_error("sphere buffer too small");
            return RTM_ERR_CAPACITY;
        }
        for (size_t i = 0; i < objs.size(); ++i) {
            rtm_sphere& s = spheres[i];
            std::memset(&s, 0, sizeof s);
            for (int k = 0; k < 3; ++k) {
                s.center[k] = objs[i].position[k];
                s.color[k] = objs[i].color[k];
                s.emission[k] = objs[i].emission[k];
            }
            s.radius = objs[i].size;
        }
    }
    return RTM_OK;
}

static int read_file(const char* path, std::string& text) {
    if (!path) {
        set_last_error("null path");
        return RTM_ERR_INVALID_ARGUMENT;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        set_last_error(std::string("cannot open ") + path);
        return RTM_ERR_IO;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    text = ss.str();
    return RTM_OK;
}

int scene_load_json_objects(const char* path, int literal_loader, rtm_settings* st, rtm_object* objects,
                            size_t capacity, size_t* n_objects) {
    std::string text;
    const int rc = read_file(path, text);
    if (rc != RTM_OK) return rc;
    return scene_parse_json_objects(text.data(), text.size(), literal_loader, st, objects, capacity, n_objects);
}

int scene_load_json(const char* path, int literal_loader, rtm_settings* st, rtm_sphere* spheres,
                    size_t capacity, size_t* n_spheres) {
    std::string text;
    const int rc = read_file(path, text);
    if (rc != RTM_OK) return rc;
    return scene_parse_json(text.data(), text.size(), literal_loader, st, spheres, capacity, n_spheres);
}

// LoadData::SaveSampleJson (src/SettingData.cpp:14-24,100-103) through to_json (:106-118): the
// object list is not serialised (Q22); nlohmann dumps keys sorted, compact.
int scene_save_sample_json(const char* path) {
    if (!path) return RTM_ERR_INVALID_ARGUMENT;
    std::ofstream f(path, std::ios::binary);
    if (!f) {
        set_last_error(std::string("cannot write ") + path);
        return RTM_ERR_IO;
    }
    f << "{\"00 height\":540,\"00 samples\":10,\"00 superSamples\":4,\"00 width\":960,"
         "\"01 camera\":{\"fov\":60.0,\"origin\":[0.0,0.0,0.0],\"target\":[0.0,0.0,1.0],"
         "\"upVec\":[0.0,1.0,0.0]}}";
    return f.good() ? RTM_OK : RTM_ERR_IO;
}

// BASELINE config 5: SplitMix64 stream, draw order cx, cy, cz, radius,
// colour r, g, b per sphere; every 50th sphere is a light.
int scene_make_stress(uint64_t seed, size_t n, rtm_settings* st, rtm_sphere* spheres) {
    if (!st || (!spheres && n)) return RTM_ERR_INVALID_ARGUMENT;
    uint64_t state = seed;
    auto next_u = [&state]() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return (double)(z >> 11) * (1.0 / 9007199254740992.0);
    };
    for (size_t i = 0; i < n; ++i) {
        rtm_sphere& s = spheres[i];
        std::memset(&s, 0, sizeof s);
        for (int k = 0; k < 3; ++k) s.center[k] = -50.0 + 100.0 * next_u();
        s.radius = (float)(0.2 + 0.8 * next_u());
        for (int k = 0; k < 3; ++k) s.color[k] = 0.1 + 0.8 * next_u();
        const double e = (i % 50 == 0) ? 5.0 : 0.0;
        s.emission[0] = s.emission[1] = s.emission[2] = e;
    }
    std::memset(st, 0, sizeof *st);
    st->width = 1920;
    st->height = 1080;
    st->samples = 256;
    st->super_samples = 1;
    st->camera.origin[2] = -60.0;
    st->camera.up[1] = 1.0;
    st->camera.fov = 1.0f;
    return RTM_OK;
}

}  // namespace rtm
