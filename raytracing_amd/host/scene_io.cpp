// scene_io.cpp — see scene_io.hpp.  Written from the behaviour of Sandbox/WavefrontLoader.cpp and
// Sandbox/Sandbox.cpp:578-763, not from their text: a hand-rolled tokenizer and a small expression evaluator.
#include "scene_io.hpp"

#include <cctype>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <sstream>

namespace sbio {

namespace {
[[noreturn]] void Fail(const std::string &where, size_t line, const std::string &what) {
    throw std::runtime_error(where + " line " + std::to_string(line) + ": " + what);
}
std::vector<std::string> SplitBlanks(const std::string &s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !std::isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}
}  // namespace

WfData LoadWavefront(const std::string &fileName) {
    std::ifstream is(fileName);
    if (!is) throw std::runtime_error("cannot open " + fileName);
    WfData result;
    std::string group = "default";
    std::vector<re::Vector3> vertices, normals;
    std::vector<re::Vector2> texCoords;
    std::string line;
    size_t lineNo = 0;
    auto number = [&](const std::vector<std::string> &tok, size_t k) -> re::real {
        return k < tok.size() ? std::strtod(tok[k].c_str(), nullptr) : 0.0;  // a short line leaves the stream's 0
    };
    while (std::getline(is, line)) {
        lineNo++;
        const std::vector<std::string> tok = SplitBlanks(line);
        if (tok.empty()) continue;
        if (tok[0] == "v") {
            vertices.push_back({number(tok, 1), number(tok, 2), number(tok, 3)});
        } else if (tok[0] == "vt") {
            texCoords.push_back({number(tok, 1), number(tok, 2)});
        } else if (tok[0] == "vn") {
            normals.push_back({number(tok, 1), number(tok, 2), number(tok, 3)});
        } else if (tok[0] == "g") {
            if (tok.size() > 1) group = tok[1];
        } else if (tok[0] == "f") {
            WfFace face;
            for (size_t k = 1; k < tok.size(); k++) {
                // v | v/vt | v//vn | v/vt/vn, 1-based; a missing field is index 0 -> -1 -> "none"
                int idx[3] = {0, 0, 0};
                size_t field = 0, start = 0;
                const std::string &s = tok[k];
                for (size_t p = 0; p <= s.size() && field < 3; p++) {
                    if (p == s.size() || s[p] == '/') {
                        idx[field++] = std::atoi(s.substr(start, p - start).c_str());
                        start = p + 1;
                    }
                }
                const int v = idx[0] - 1, vt = idx[1] - 1, vn = idx[2] - 1;
                if (v < 0 || (size_t)v >= vertices.size()) Fail(fileName, lineNo, "vertex index out of range in '" + s + "'");
                if (vt >= (int)texCoords.size()) Fail(fileName, lineNo, "texture index out of range in '" + s + "'");
                if (vn >= (int)normals.size()) Fail(fileName, lineNo, "normal index out of range in '" + s + "'");
                face.Vertices.push_back(vertices[v]);
                face.TexCoords.push_back(vt >= 0 ? texCoords[vt] : re::Vector2{0, 0});
                face.Normals.push_back(vn >= 0 ? normals[vn] : re::Vector3::Zero);
            }
            result[group].push_back(face);
        }
    }
    return result;
}

// ------------------------------------------------------------------------------------------- scene scripts
namespace {
struct Value {
    bool isString = false;
    double num = 0;
    std::string str;
};
struct Token {
    enum Kind { End, Name, Number, String, Punct } kind = End;
    std::string text;
    double num = 0;
    size_t line = 1;
};

class Lexer {
public:
    explicit Lexer(const std::string &s) : m_S(s) { Next(); }
    const Token &Peek() const { return m_Tok; }
    Token Take() { Token t = m_Tok; Next(); return t; }
private:
    void Next() {
        for (;;) {  // blanks and comments
            while (m_P < m_S.size() && std::isspace((unsigned char)m_S[m_P])) { if (m_S[m_P] == '\n') m_Line++; m_P++; }
            if (m_P + 1 < m_S.size() && m_S[m_P] == '-' && m_S[m_P + 1] == '-') {
                if (m_S.compare(m_P, 4, "--[[") == 0) {
                    size_t e = m_S.find("]]", m_P + 4);
                    for (size_t k = m_P; k < std::min(e, m_S.size()); k++) if (m_S[k] == '\n') m_Line++;
                    m_P = e == std::string::npos ? m_S.size() : e + 2;
                } else {
                    while (m_P < m_S.size() && m_S[m_P] != '\n') m_P++;
                }
                continue;
            }
            break;
        }
        m_Tok = Token();
        m_Tok.line = m_Line;
        if (m_P >= m_S.size()) return;
        const char c = m_S[m_P];
        if (std::isalpha((unsigned char)c) || c == '_') {
            size_t e = m_P;
            while (e < m_S.size() && (std::isalnum((unsigned char)m_S[e]) || m_S[e] == '_')) e++;
            m_Tok.kind = Token::Name;
            m_Tok.text = m_S.substr(m_P, e - m_P);
            m_P = e;
        } else if (std::isdigit((unsigned char)c) || (c == '.' && m_P + 1 < m_S.size() && std::isdigit((unsigned char)m_S[m_P + 1]))) {
            char *end = nullptr;
            m_Tok.kind = Token::Number;
            if (c == '0' && m_P + 1 < m_S.size() && (m_S[m_P + 1] == 'x' || m_S[m_P + 1] == 'X'))
                m_Tok.num = (double)std::strtoull(m_S.c_str() + m_P, &end, 16);
            else
                m_Tok.num = std::strtod(m_S.c_str() + m_P, &end);
            m_P = (size_t)(end - m_S.c_str());
        } else if (c == '"' || c == '\'') {
            size_t e = m_P + 1;
            std::string out;
            while (e < m_S.size() && m_S[e] != c) {
                if (m_S[e] == '\\' && e + 1 < m_S.size()) e++;
                out += m_S[e++];
            }
            if (e >= m_S.size()) Fail("script", m_Line, "unfinished string");
            m_Tok.kind = Token::String;
            m_Tok.text = out;
            m_P = e + 1;
        } else {
            m_Tok.kind = Token::Punct;
            m_Tok.text = std::string(1, c);
            m_P++;
        }
    }
    const std::string &m_S;
    size_t m_P = 0, m_Line = 1;
    Token m_Tok;
};

class Interpreter {
public:
    Interpreter(const std::string &resDir, ScriptScene &out) : m_ResDir(resDir), m_Out(out) {
        m_Out = ScriptScene();
        m_Out.Scene = std::make_shared<re::Scene>();
        Register();
    }
    void Run(const std::string &code) {
        Lexer lx(code);
        while (lx.Peek().kind != Token::End) {
            if (lx.Peek().kind == Token::Punct && lx.Peek().text == ";") { lx.Take(); continue; }
            if (lx.Peek().kind != Token::Name) Fail("script", lx.Peek().line, "unexpected symbol near '" + lx.Peek().text + "'");
            Token name = lx.Take();
            if (name.text == "local") {
                if (lx.Peek().kind != Token::Name) Fail("script", name.line, "<name> expected near 'local'");
                name = lx.Take();
            }
            if (lx.Peek().kind == Token::Punct && lx.Peek().text == "=") {
                lx.Take();
                m_Globals[name.text] = Expr(lx);
            } else if (lx.Peek().kind == Token::Punct && lx.Peek().text == "(") {
                Call(name, lx);
            } else {
                Fail("script", name.line, "syntax error near '" + name.text + "'");
            }
        }
    }
private:
    using Fn = std::function<Value(const std::vector<Value> &, size_t line)>;
    Value Expr(Lexer &lx) {
        const Token t = lx.Take();
        switch (t.kind) {
        case Token::Number: { Value v; v.num = t.num; return v; }
        case Token::String: { Value v; v.isString = true; v.str = t.text; return v; }
        case Token::Name:
            if (lx.Peek().kind == Token::Punct && lx.Peek().text == "(") return Call(t, lx);
            if (!m_Globals.count(t.text)) Fail("script", t.line, "variable '" + t.text + "' is nil");
            return m_Globals[t.text];
        case Token::Punct:
            if (t.text == "-") { Value v = Expr(lx); if (v.isString) Fail("script", t.line, "attempt to negate a string"); v.num = -v.num; return v; }
            if (t.text == "(") { Value v = Expr(lx); Expect(lx, ")"); return v; }
            [[fallthrough]];
        default:
            Fail("script", t.line, "unexpected symbol near '" + t.text + "'");
        }
    }
    void Expect(Lexer &lx, const char *p) {
        if (lx.Peek().kind != Token::Punct || lx.Peek().text != p) Fail("script", lx.Peek().line, std::string("'") + p + "' expected near '" + lx.Peek().text + "'");
        lx.Take();
    }
    Value Call(const Token &name, Lexer &lx) {
        Expect(lx, "(");
        std::vector<Value> args;
        if (!(lx.Peek().kind == Token::Punct && lx.Peek().text == ")")) {
            for (;;) {
                args.push_back(Expr(lx));
                if (lx.Peek().kind == Token::Punct && lx.Peek().text == ",") { lx.Take(); continue; }
                break;
            }
        }
        Expect(lx, ")");
        auto it = m_Fns.find(name.text);
        if (it == m_Fns.end()) Fail("script", name.line, "attempt to call a nil value (global '" + name.text + "')");
        if (args.size() != it->second.first) Fail("script", name.line, name.text + " expects " + std::to_string(it->second.first) + " argument(s)");
        return it->second.second(args, name.line);
    }
    static Value Num(double d) { Value v; v.num = d; return v; }
    double N(const std::vector<Value> &a, size_t k, size_t line) const {
        if (a[k].isString) Fail("script", line, "number expected, got string");
        return a[k].num;
    }
    template <typename T> size_t Index(const std::vector<T> &v, double d, const char *what, size_t line) const {
        const long long i = (long long)d;
        if (i < 0 || (size_t)i >= v.size()) Fail("script", line, std::string("Invalid ") + what + ": " + std::to_string(i));  // Sandbox.cpp CheckSize
        return (size_t)i;
    }
    re::SceneNode *NewShapeNode() {  // Sandbox.cpp:650-655: a child of the root with the sticky position / scale
        auto node = m_Out.Scene->GetRoot()->AddChild();
        auto tr = node->GetComponentOfType<re::Transform>();
        tr->Position = m_Position;
        tr->Scale = m_Scale;
        m_Out.ShapeNodes.push_back(node.get());
        return node.get();
    }
    void Def(const char *name, size_t arity, Fn fn) { m_Fns[name] = {arity, std::move(fn)}; }
    void Register() {
        auto &o = m_Out;
        Def("reUniformMaterial", 2, [&](const std::vector<Value> &a, size_t l) {
            o.Materials.push_back(std::make_shared<re::UniformMaterial>(re::Color((unsigned int)(long long)N(a, 0, l)), N(a, 1, l)));
            return Num((double)o.Materials.size() - 1);
        });
        Def("reInterpolatedMaterial", 3, [&](const std::vector<Value> &a, size_t l) {
            const size_t m0 = Index(o.Materials, N(a, 1, l), "material", l), m1 = Index(o.Materials, N(a, 2, l), "material", l);
            const size_t n = Index(o.Noises, N(a, 0, l), "noise", l);
            o.Materials.push_back(std::make_shared<re::InterpolatedMaterial>(o.Noises[n], o.Materials[m0], o.Materials[m1]));
            return Num((double)o.Materials.size() - 1);
        });
        Def("reCheckerBoard", 1, [&](const std::vector<Value> &a, size_t l) {
            o.Noises.push_back(std::make_shared<re::CheckerBoard>(N(a, 0, l)));
            return Num((double)o.Noises.size() - 1);
        });
        Def("rePerlin", 1, [&](const std::vector<Value> &a, size_t l) {
            o.Noises.push_back(std::make_shared<re::Perlin>(N(a, 0, l)));
            return Num((double)o.Noises.size() - 1);
        });
        Def("reWorley", 2, [&](const std::vector<Value> &a, size_t l) {
            o.Noises.push_back(std::make_shared<re::Worley>(N(a, 0, l), (int)N(a, 1, l)));
            return Num((double)o.Noises.size() - 1);
        });
        Def("reMarble", 3, [&](const std::vector<Value> &a, size_t l) {
            o.Noises.push_back(std::make_shared<re::Marble>(N(a, 0, l), N(a, 1, l), N(a, 2, l)));
            return Num((double)o.Noises.size() - 1);
        });
        Def("rePosition", 3, [&](const std::vector<Value> &a, size_t l) { m_Position = {N(a, 0, l), N(a, 1, l), N(a, 2, l)}; return Value(); });
        Def("reScale", 3, [&](const std::vector<Value> &a, size_t l) { m_Scale = {N(a, 0, l), N(a, 1, l), N(a, 2, l)}; return Value(); });
        Def("reSphere", 1, [&](const std::vector<Value> &a, size_t l) {
            const size_t m = Index(o.Materials, N(a, 0, l), "material", l);
            auto sphere = NewShapeNode()->AddComponent<re::Sphere>();
            sphere->Material = o.Materials[m].get();
            return Value();
        });
        Def("rePlane", 4, [&](const std::vector<Value> &a, size_t l) {
            const size_t m = Index(o.Materials, N(a, 0, l), "material", l);
            auto plane = NewShapeNode()->AddComponent<re::Plane>();
            plane->Normal = {N(a, 1, l), N(a, 2, l), N(a, 3, l)};  // not normalised (Sandbox.cpp:664)
            plane->Material = o.Materials[m].get();
            return Value();
        });
        Def("reObjMesh", 3, [&](const std::vector<Value> &a, size_t l) {
            const size_t m = Index(o.Materials, N(a, 0, l), "material", l);
            if (!a[1].isString || !a[2].isString) Fail("script", l, "reObjMesh(material, \"file\", \"group\")");
            std::string file = a[1].str;
            if (!m_ResDir.empty() && !file.empty() && file[0] != '/') file = m_ResDir + "/" + file;
            const WfData data = LoadWavefront(file);
            auto mesh = NewShapeNode()->AddComponent<re::Mesh>();
            auto g = data.find(a[2].str);
            if (g == data.end()) Fail("script", l, "Invalid obj group: " + a[2].str);
            for (const WfFace &f : g->second) {  // the first three vertices of every face (Sandbox.cpp:683-688)
                if (f.Vertices.size() < 3) Fail("script", l, "obj face with fewer than 3 vertices");
                re::Triangle &t = mesh->AddTriangle();
                t.Vertices = {f.Vertices[0], f.Vertices[1], f.Vertices[2]};
                t.Normals = {f.Normals[0], f.Normals[1], f.Normals[2]};
            }
            mesh->NormalMode = re::NormalModes::Vertex;
            mesh->Material = o.Materials[m].get();
            return Value();
        });
        auto addLight = [&](std::shared_ptr<re::Light> light) {
            o.Scene->Lights.push_back(light);
            o.Lights.push_back(light);
            return Num((double)o.Lights.size() - 1);
        };
        Def("reAmbientLight", 1, [&, addLight](const std::vector<Value> &a, size_t l) {
            auto light = std::make_shared<re::Light>();
            light->Type = re::LightType::Ambient;
            light->Color = re::Color((unsigned int)(long long)N(a, 0, l));
            return addLight(light);
        });
        Def("reDirectionalLight", 4, [&, addLight](const std::vector<Value> &a, size_t l) {
            auto light = std::make_shared<re::Light>();
            light->Type = re::LightType::Directional;
            light->Color = re::Color((unsigned int)(long long)N(a, 0, l));
            light->Direction = re::Vector3(N(a, 1, l), N(a, 2, l), N(a, 3, l)).Normalized();
            return addLight(light);
        });
        Def("rePointLight", 2, [&, addLight](const std::vector<Value> &a, size_t l) {
            auto light = std::make_shared<re::Light>();
            light->Type = re::LightType::Point;
            light->Color = re::Color((unsigned int)(long long)N(a, 0, l));
            light->Position = m_Position;
            light->Attenuation = N(a, 1, l);
            return addLight(light);
        });
        Def("reCameraPos", 3, [&](const std::vector<Value> &a, size_t l) {
            o.Scene->Camera.Position = {N(a, 0, l), N(a, 1, l), N(a, 2, l)};
            return Value();
        });
        Def("reSkyBox", 3, [&](const std::vector<Value> &a, size_t l) {
            const size_t sun = Index(o.Lights, N(a, 2, l), "light", l);
            o.Scene->Background = std::make_shared<re::SkyBox>(re::Color((unsigned int)(long long)N(a, 0, l)),
                                                               re::Color((unsigned int)(long long)N(a, 1, l)), o.Lights[sun]);
            return Value();
        });
    }
    std::string m_ResDir;
    ScriptScene &m_Out;
    re::Vector3 m_Position = re::Vector3::Zero, m_Scale = {1, 1, 1};
    std::map<std::string, Value> m_Globals;
    std::map<std::string, std::pair<size_t, Fn>> m_Fns;
};
}  // namespace

void RunSceneScript(const std::string &code, const std::string &resDir, ScriptScene &out) {
    Interpreter in(resDir, out);
    in.Run(code);
}

std::vector<std::pair<std::string, std::string>> ReadScenesXml(const std::string &path) {
    std::ifstream is(path, std::ios::binary);
    if (!is) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << is.rdbuf();
    const std::string s = ss.str();
    auto unescape = [](const std::string &t) {
        static const std::pair<const char *, char> ent[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
        std::string out;
        for (size_t i = 0; i < t.size();) {
            bool hit = false;
            if (t[i] == '&')
                for (auto &e : ent)
                    if (t.compare(i, std::char_traits<char>::length(e.first), e.first) == 0) {
                        out += e.second;
                        i += std::char_traits<char>::length(e.first);
                        hit = true;
                        break;
                    }
            if (!hit) out += t[i++];
        }
        return out;
    };
    std::vector<std::pair<std::string, std::string>> scenes;
    size_t p = 0;
    while ((p = s.find("<scene", p)) != std::string::npos) {
        if (s.compare(p, 7, "<scenes") == 0) { p += 7; continue; }
        const size_t close = s.find('>', p);
        if (close == std::string::npos) break;
        std::string name;
        const size_t n0 = s.find("name=\"", p);
        if (n0 != std::string::npos && n0 < close) {
            const size_t n1 = s.find('"', n0 + 6);
            name = unescape(s.substr(n0 + 6, n1 - n0 - 6));
        }
        const size_t end = s.find("</scene>", close);
        if (end == std::string::npos) throw std::runtime_error(path + ": <scene> without </scene>");
        scenes.emplace_back(name, unescape(s.substr(close + 1, end - close - 1)));
        p = end + 8;
    }
    return scenes;
}

void WritePPM(const std::string &path, const unsigned int *pixels, size_t width, size_t height) {
    std::ofstream os(path, std::ios::binary);
    if (!os) throw std::runtime_error("cannot write " + path);
    os << "P6\n" << width << " " << height << "\n255\n";
    std::vector<unsigned char> row(3 * width);
    for (size_t y = 0; y < height; y++) {
        for (size_t x = 0; x < width; x++) {
            const unsigned int v = pixels[y * width + x];
            row[3 * x] = v & 0xff;
            row[3 * x + 1] = (v >> 8) & 0xff;
            row[3 * x + 2] = (v >> 16) & 0xff;
        }
        os.write(reinterpret_cast<const char *>(row.data()), (std::streamsize)row.size());
    }
}

}  // namespace sbio
