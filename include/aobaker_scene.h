// aobaker_scene.h — headless stand-ins for the reference's scene classes, reduced to
// exactly what BakeEngine::Bake reads.
//
// The reference's own classes are bound to Direct3D 11, Dear ImGui and assimp
// (Scene/Scene.h, Scene/Objects/BakeObject.h, Render/RenderMesh.h, ...), none of which
// exist on an MI355X host.  A maintainer of the reference keeps their classes and only
// re-points BakeEngine.cpp at aobaker.h; a headless caller uses these.
// Class, method and field names are the reference's, so code written against
// `Scene::GetObjectCount()`, `Section::GetPreviewMesh()`, `BakeTexture::GetData()` ...
// compiles unchanged.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace DirectX { namespace SimpleMath {
struct Vector2 { float x = 0, y = 0; Vector2() = default; Vector2(float x_, float y_) : x(x_), y(y_) {} };
struct Vector3 { float x = 0, y = 0, z = 0; Vector3() = default; Vector3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {} };
struct Vector4 { float x = 0, y = 0, z = 0, w = 0; Vector4() = default; Vector4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {} };
// row-major, row-vector convention (v' = v * M), as SimpleMath::Matrix
struct Matrix {
    float m[4][4];
    Matrix() { std::memset(m, 0, sizeof m); m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1.0f; }
    Matrix operator*(const Matrix& o) const {
        Matrix r;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            float s = 0.0f;
            for (int k = 0; k < 4; k++) s += m[i][k] * o.m[k][j];
            r.m[i][j] = s;
        }
        return r;
    }
    static Matrix CreateScale(const Vector3& s) { Matrix r; r.m[0][0] = s.x; r.m[1][1] = s.y; r.m[2][2] = s.z; return r; }
    static Matrix CreateTranslation(const Vector3& t) { Matrix r; r.m[3][0] = t.x; r.m[3][1] = t.y; r.m[3][2] = t.z; return r; }
    // XMQuaternionRotationRollPitchYaw(pitch, yaw, roll) -> XMMatrixRotationQuaternion
    // (SimpleMath.inl:3072-3078, 2596-2603)
    static Matrix CreateFromYawPitchRoll(float yaw, float pitch, float roll) {
        const float hp = pitch * 0.5f, hy = yaw * 0.5f, hr = roll * 0.5f;
        const float sp = std::sin(hp), cp = std::cos(hp), sy = std::sin(hy), cy = std::cos(hy), sr = std::sin(hr), cr = std::cos(hr);
        const float x = cr * sp * cy + sr * cp * sy, y = cr * cp * sy - sr * sp * cy;
        const float z = sr * cp * cy - cr * sp * sy, w = cr * cp * cy + sr * sp * sy;
        Matrix r;
        r.m[0][0] = 1 - 2 * y * y - 2 * z * z; r.m[0][1] = 2 * x * y + 2 * z * w;     r.m[0][2] = 2 * x * z - 2 * y * w;
        r.m[1][0] = 2 * x * y - 2 * z * w;     r.m[1][1] = 1 - 2 * x * x - 2 * z * z; r.m[1][2] = 2 * y * z + 2 * x * w;
        r.m[2][0] = 2 * x * z + 2 * y * w;     r.m[2][1] = 2 * y * z - 2 * x * w;     r.m[2][2] = 1 - 2 * x * x - 2 * y * y;
        return r;
    }
};
}} // namespace DirectX::SimpleMath

// Render/RenderMesh.h:13-23 — 48 bytes: pos@0, normal@12, uv@24, color@32
struct RenderVertex {
    DirectX::SimpleMath::Vector3 Position;
    DirectX::SimpleMath::Vector3 Normal;
    DirectX::SimpleMath::Vector2 UV;
    DirectX::SimpleMath::Vector4 Color;
};
static_assert(sizeof(RenderVertex) == 48, "RenderVertex must match the reference's 48-byte layout");

// Render/RenderMesh.cpp:29-45 keeps CPU copies of the vertex and index arrays; Bake reads those.
class RenderMesh {
public:
    RenderMesh(const RenderVertex* vertices, int vertexCount, const unsigned int* indices, int indexCount)
        : Vertices(vertices, vertices + vertexCount), Indices(indices, indices + indexCount) {}
    int GetVertexCount() const { return (int)Vertices.size(); }
    RenderVertex* GetVertices() const { return const_cast<RenderVertex*>(Vertices.data()); }
    int GetIndexCount() const { return (int)Indices.size(); }
    unsigned int* GetIndices() const { return const_cast<unsigned int*>(Indices.data()); }
private:
    std::vector<RenderVertex> Vertices;
    std::vector<unsigned int> Indices;
};

// Scene/BakeTexture.{h,cpp}: W x H x channels x bytes, tightly packed, initialised to 0xFF
class BakeTexture {
public:
    BakeTexture(int width, int height, int channelCount, int bytesPerChannel)
        : Width(width), Height(height), ChannelCount(channelCount), BytesPerChannel(bytesPerChannel),
          Data((size_t)width * height * channelCount * bytesPerChannel, (unsigned char)0xFF) {}
    int GetWidth() const { return Width; }
    int GetHeight() const { return Height; }
    int GetChannelCount() const { return ChannelCount; }
    int GetBytesPerChannel() const { return BytesPerChannel; }
    unsigned char* GetData() const { return const_cast<unsigned char*>(Data.data()); }
    size_t GetDataLength() const { return Data.size(); }
    void RefreshPreview() const {}          // GPU preview upload (BakeTexture.cpp:119-122): nothing to do headless
    // BakeTexture.cpp:104-116 (stb_image_write there; 8-bit only).  Here: 8-bit PNG for 1-byte channels, 16-bit PNG
    // (UNORM of the float value clamped to [0,1]) for the 2- and 4-byte float channels.  aobaker_amd/host/Assets.cpp
    bool SavePNG(const char* filename) const;
    void SavePNG(const wchar_t* filename) const;
    // not in the reference: the image at full precision — an uncompressed scan-line OpenEXR (FLOAT, or HALF for 2-byte channels)
    bool SaveEXR(const char* filename) const;
private:
    int Width, Height, ChannelCount, BytesPerChannel;
    std::vector<unsigned char> Data;
};

class MaterialObject {
public:
    // MaterialObject.cpp:7 hard-codes BakeTexture(512, 512, 1, 1); here the atlas is a parameter
    explicit MaterialObject(const std::wstring& name, int width = 512, int height = 512, int channelCount = 1, int bytesPerChannel = 1)
        : Name(name), AOTexture(new BakeTexture(width, height, channelCount, bytesPerChannel)) {}
    BakeTexture* GetAOTexture() const { return AOTexture.get(); }
    const std::wstring& GetName() const { return Name; }
    // MaterialObject.h:10-11, default 1.0f (MaterialObject.cpp:4); Bake reads it only with AobOptions::material_ranges
    float GetAOMaxRange() const { return AOMaxRange; }
    void SetAOMaxRange(const float& aoMaxRange) { AOMaxRange = aoMaxRange; }
private:
    std::wstring Name;
    std::unique_ptr<BakeTexture> AOTexture;
    float AOMaxRange = 1.0f;
};

// Scene/Objects/TransformObject.cpp:16-24
class TransformObject {
public:
    DirectX::SimpleMath::Vector3 GetPosition() const { return Position; }
    void SetPosition(const DirectX::SimpleMath::Vector3& v) { Position = v; }
    DirectX::SimpleMath::Vector3 GetRotation() const { return Rotation; }
    void SetRotation(const DirectX::SimpleMath::Vector3& v) { Rotation = v; }
    DirectX::SimpleMath::Vector3 GetScale() const { return Scale; }
    void SetScale(const DirectX::SimpleMath::Vector3& v) { Scale = v; }
    DirectX::SimpleMath::Matrix BuildTransform() const {
        using DirectX::SimpleMath::Matrix;
        return Matrix::CreateScale(Scale) * Matrix::CreateFromYawPitchRoll(Rotation.y, Rotation.x, Rotation.z) * Matrix::CreateTranslation(Position);
    }
private:
    DirectX::SimpleMath::Vector3 Position, Rotation, Scale { 1.0f, 1.0f, 1.0f };
};

class Scene;

class BakeObject : public TransformObject {
public:
    class Section {
    public:
        Section(const std::wstring& name, RenderMesh* mesh, MaterialObject* material) : Name(name), PreviewMesh(mesh), Material(material) {}
        MaterialObject* GetMaterial() const { return Material; }
        void SetMaterial(MaterialObject* m) { Material = m; }
        const std::wstring& GetName() const { return Name; }
        RenderMesh* GetPreviewMesh() const { return PreviewMesh.get(); }
        // BakeObject.h:28-31, both default to true (BakeObject.cpp:20)
        bool IsVisibleToSelf() const { return VisibleToSelf; }
        void SetVisibleToSelf(const bool& value) { VisibleToSelf = value; }
        bool IsVisibleToOthers() const { return VisibleToOthers; }
        void SetVisibleToOthers(const bool& value) { VisibleToOthers = value; }
    private:
        std::wstring Name;
        std::unique_ptr<RenderMesh> PreviewMesh;
        MaterialObject* Material;
        bool VisibleToSelf = true, VisibleToOthers = true;
    };
    explicit BakeObject(const std::wstring& name) : Name(name) {}
    // BakeObject.cpp:170-195 (assimp there).  Wavefront OBJ only, with the reference's import conventions:
    // triangulated, normals generated where absent, left-handed with V flipped, one section per object/material;
    // materials are looked up in / added to the target scene by name.  New materials get an atlasWidth x atlasHeight
    // AO texture (MaterialObject.cpp:7 hard-codes 512 x 512 x 1 x 1).  aobaker_amd/host/Assets.cpp
    bool LoadFromFile(const char* filename, Scene* const targetScene, int atlasWidth = 512, int atlasHeight = 512,
                      int channelCount = 1, int bytesPerChannel = 1);
    void LoadFromFile(const std::wstring& filename, Scene* const targetScene);
    const std::wstring& GetName() const { return Name; }
    Section* AddSection(Section* s) { Sections.emplace_back(s); return s; }     // takes ownership
    int GetSectionCount() const { return (int)Sections.size(); }
    Section* GetSection(int index) const { return Sections[(size_t)index].get(); }
private:
    std::wstring Name;
    std::vector<std::unique_ptr<Section>> Sections;
};

// Scene/Scene.h:21-32 — owns its objects and materials (Scene.cpp:248-254)
class Scene {
public:
    void AddObject(BakeObject* object) { BakeObjects.emplace_back(object); }
    BakeObject* GetObject(const int& index) const { return BakeObjects[(size_t)index].get(); }
    int GetObjectCount() const { return (int)BakeObjects.size(); }
    void AddMaterial(MaterialObject* material) { Materials.emplace_back(material); }
    MaterialObject* GetMaterial(const int& index) const { return Materials[(size_t)index].get(); }
    int GetMaterialCount() const { return (int)Materials.size(); }
    int FindMaterial(MaterialObject* material) const {
        for (size_t i = 0; i < Materials.size(); i++) if (Materials[i].get() == material) return (int)i;
        return -1;
    }
    int FindMaterial(const std::wstring& name) const {                         // Scene.h:29
        for (size_t i = 0; i < Materials.size(); i++) if (Materials[i]->GetName() == name) return (int)i;
        return -1;
    }
private:
    std::vector<std::unique_ptr<BakeObject>> BakeObjects;
    std::vector<std::unique_ptr<MaterialObject>> Materials;
};
