// compv_hip_plugin.cxx -- the reference-side binding: CompV C++ classes that implement CompV's own abstract
// CompVEdgeDete / CompVHough interfaces on top of the C ABI in include/compv_hip.h, and the factory table that
// re-registers the ids COMPV_CANNY_ID / COMPV_SOBEL_ID / COMPV_SCHARR_ID / COMPV_PREWITT_ID / COMPV_HOUGHSHT_ID / COMPV_HOUGHKHT_ID.
//
// CompVFeature::addFactory() REPLACES an existing id (base/compv_features.cxx:30-40), so after
//     CompVInit(); compv_hip_plugin_register();
// every CompVEdgeDete::newObj(&p, COMPV_CANNY_ID, ...) / CompVHough::newObj(&h, COMPV_HOUGHSHT_ID, ...) in
// samples/edges_canny, samples/hough_lines and unittests/*.cxx returns the HIP implementation, unchanged call sites.
//
// This file is compiled against a CompV checkout (-I$COMPV_ROOT/base/include ...).  It derives from the public abstract classes
// (base/include/compv/base/compv_features.h:207-227) and keeps their CONTRACT -- option ids, value sizes, accepted ranges, error
// codes (core/features/edges/compv_core_feature_canny_dete.cxx:77-117, core/features/hough/compv_core_feature_houghsht.cxx:60-94,
// core/features/hough/compv_core_feature_houghkht.cxx:140-206) -- but none of the reference's code: options are described by a
// table and applied by one routine, and every computation (toCartesian included) lives behind the C ABI.
#include <compv/base/compv_base.h>
#include <compv/base/compv_features.h>
#include <compv/base/compv_debug.h>
#include <compv/base/image/compv_image.h>

#include "../include/compv_hip.h"

#include <climits>
#include <cstring>
#include <vector>

#define COMPV_THIS_CLASSNAME "CompVHipPlugin"

COMPV_NAMESPACE_BEGIN()

namespace hipbinding {

// COMPVHIP_E_* -> COMPV_ERROR_CODE (table in INTEGRATION.md)
static COMPV_ERROR_CODE translate(int rc)
{
	static const struct { int hip; COMPV_ERROR_CODE compv; } map[] = {
		{ COMPVHIP_OK, COMPV_ERROR_CODE_S_OK },
		{ COMPVHIP_E_NOT_IMPLEMENTED, COMPV_ERROR_CODE_E_NOT_IMPLEMENTED },
		{ COMPVHIP_E_NOT_INITIALIZED, COMPV_ERROR_CODE_E_NOT_INITIALIZED },
		{ COMPVHIP_E_INVALID_STATE, COMPV_ERROR_CODE_E_INVALID_STATE },
		{ COMPVHIP_E_INVALID_PARAMETER, COMPV_ERROR_CODE_E_INVALID_PARAMETER },
		{ COMPVHIP_E_OUT_OF_MEMORY, COMPV_ERROR_CODE_E_OUT_OF_MEMORY },
		{ COMPVHIP_E_OUT_OF_BOUND, COMPV_ERROR_CODE_E_OUT_OF_BOUND },
	};
	for (size_t i = 0; i < sizeof(map) / sizeof(map[0]); ++i) if (map[i].hip == rc) return map[i].compv;
	return COMPV_ERROR_CODE_E_CUDA; // no HIP code exists in CompV; E_CUDA is the GPU-runtime class (compv_common.h:266)
}

static COMPV_ERROR_CODE report(compvhip_ctx* ctx, const char* what, int rc)
{
	if (rc != COMPVHIP_OK) COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "%s failed (%d): %s", what, rc, compvhip_last_error(ctx));
	return translate(rc);
}

// One compvhip_ctx per detector object, created on first use: a CompV detector instance owns its scratch and is not
// re-entrant (core/features/edges/compv_core_feature_canny_dete.cxx:133-147); the same holds here.
class LazyCtx
{
public:
	LazyCtx() : m_p(NULL) { }
	~LazyCtx() { compvhip_ctx_destroy(m_p); }
	COMPV_ERROR_CODE acquire(compvhip_ctx*& ctx)
	{
		if (!m_p) {
			const int rc = compvhip_ctx_create(&m_p, -1);
			if (rc != COMPVHIP_OK) {
				COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "no usable HIP device (%d): the HIP plugin has no CPU fallback", rc);
				return translate(rc);
			}
		}
		ctx = m_p;
		return COMPV_ERROR_CODE_S_OK;
	}
private:
	compvhip_ctx* m_p;
};

// ---- options ---------------------------------------------------------------------------------------------------
// What a set(id, ptr, size) call may change, as data.  `kind` fixes the value size the reference insists on; a value is accepted when
// v > 0 (FLT_POS, INT_POS), 0 < v <= hi (FLT_RANGE), v >= 0 (FLT_NONNEG), v in `allowed` (INT_ENUM2), anything (FLT_ANY / INT_ANY / BOOL_ANY).
enum OptKind { FLT_POS, FLT_RANGE, FLT_NONNEG, FLT_ANY, INT_POS, INT_ANY, INT_ENUM2, BOOL_ANY };
struct Settings { // every tunable of the four detector families; a family only lists the ones it accepts
	float cannyLow, cannyHigh; int cannyType; int kernelSize;
	float rho, thetaDeg; int threshold; int maxLines;          // maxLines <= 0: unlimited
	float clusterMinDeviation; int clusterMinSize; float kernelMinHeight;
};
struct OptSpec { int id; OptKind kind; size_t offset; float hi; int allowed[2]; };
#define OPT_AT(member) offsetof(Settings, member)

static COMPV_ERROR_CODE applyOption(const OptSpec* table, size_t count, Settings& st, int id, const void* value, size_t size, bool unknownToCaps, CompVCaps* caps)
{
	if (!value || !size) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
	for (size_t i = 0; i < count; ++i) {
		const OptSpec& o = table[i];
		if (o.id != id) continue;
		char* dst = reinterpret_cast<char*>(&st) + o.offset;
		if (o.kind == BOOL_ANY) return size == sizeof(bool) ? COMPV_ERROR_CODE_S_OK : COMPV_ERROR_CODE_E_INVALID_PARAMETER; // nothing to store
		if (o.kind == FLT_POS || o.kind == FLT_RANGE || o.kind == FLT_NONNEG || o.kind == FLT_ANY) {
			float v;
			if (size != sizeof(v)) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
			memcpy(&v, value, sizeof(v));
			const bool refused = (o.kind == FLT_NONNEG) ? (v < 0.f) : (o.kind == FLT_ANY) ? false : (v <= 0.f || (o.kind == FLT_RANGE && v > o.hi));
			if (refused) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
			memcpy(dst, &v, sizeof(v));
			return COMPV_ERROR_CODE_S_OK;
		}
		int v;
		if (size != sizeof(v)) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		memcpy(&v, value, sizeof(v));
		if (o.kind == INT_POS && v <= 0) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		if (o.kind == INT_ENUM2 && v != o.allowed[0] && v != o.allowed[1]) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		memcpy(dst, &v, sizeof(v));
		return COMPV_ERROR_CODE_S_OK;
	}
	if (unknownToCaps) return caps->CompVCaps::set(id, value, size); // the Canny class forwards unknown ids to its base (canny_dete.cxx:113)
	COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "Set with id %d not implemented", id);
	return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED; // the Hough classes refuse them (houghsht.cxx:90, houghkht.cxx:188)
}

static const OptSpec kCannyOptions[] = {
	{ COMPV_CANNY_SET_INT_THRESHOLD_TYPE, INT_ENUM2, OPT_AT(cannyType), 0.f, { COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN, COMPV_CANNY_THRESHOLD_TYPE_COMPARE_TO_GRADIENT } },
	{ COMPV_CANNY_SET_FLT32_THRESHOLD_LOW, FLT_POS, OPT_AT(cannyLow), 0.f, { 0, 0 } },
	{ COMPV_CANNY_SET_FLT32_THRESHOLD_HIGH, FLT_POS, OPT_AT(cannyHigh), 0.f, { 0, 0 } },
	{ COMPV_CANNY_SET_INT_KERNEL_SIZE, INT_ENUM2, OPT_AT(kernelSize), 0.f, { 3, 5 } },
};
static const OptSpec kShtOptions[] = {
	{ COMPV_HOUGH_SET_FLT32_RHO, FLT_RANGE, OPT_AT(rho), 1.f, { 0, 0 } },     // and must equal 1 for SHT: checked by the class
	{ COMPV_HOUGH_SET_FLT32_THETA, FLT_POS, OPT_AT(thetaDeg), 0.f, { 0, 0 } },
	{ COMPV_HOUGH_SET_INT_THRESHOLD, INT_POS, OPT_AT(threshold), 0.f, { 0, 0 } },
	{ COMPV_HOUGH_SET_INT_MAXLINES, INT_ANY, OPT_AT(maxLines), 0.f, { 0, 0 } },
};
static const OptSpec kKhtOptions[] = {
	{ COMPV_HOUGH_SET_FLT32_RHO, FLT_RANGE, OPT_AT(rho), 1.f, { 0, 0 } },
	{ COMPV_HOUGH_SET_FLT32_THETA, FLT_POS, OPT_AT(thetaDeg), 0.f, { 0, 0 } },
	{ COMPV_HOUGH_SET_INT_THRESHOLD, INT_POS, OPT_AT(threshold), 0.f, { 0, 0 } },
	{ COMPV_HOUGH_SET_INT_MAXLINES, INT_ANY, OPT_AT(maxLines), 0.f, { 0, 0 } },
	{ COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION, FLT_ANY, OPT_AT(clusterMinDeviation), 0.f, { 0, 0 } },   // the reference checks the size only
	{ COMPV_HOUGHKHT_SET_INT_CLUSTER_MIN_SIZE, INT_POS, OPT_AT(clusterMinSize), 0.f, { 0, 0 } },
	{ COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH, FLT_NONNEG, OPT_AT(kernelMinHeight), 0.f, { 0, 0 } },        // 0 keeps every kernel
	{ COMPV_HOUGHKHT_SET_BOOL_OVERRIDE_INPUT_EDGES, BOOL_ANY, 0, 0.f, { 0, 0 } }, // the HIP path always works on its own copy
};
#define OPT_COUNT(t) (sizeof(t) / sizeof((t)[0]))

static Settings defaults()
{
	Settings s;
	memset(&s, 0, sizeof(s));
	s.cannyType = COMPV_CANNY_THRESHOLD_TYPE_COMPARE_TO_GRADIENT; s.kernelSize = 3;
	s.rho = 1.f; s.thetaDeg = 1.f; s.threshold = 1; s.maxLines = 0;
	s.clusterMinDeviation = 2.0f; s.clusterMinSize = 10; s.kernelMinHeight = 0.002f; // houghkht.cxx:38-40
	return s;
}

// ---- shared pieces of process() ----------------------------------------------------------------------------------
static bool isGray8(const CompVMatPtr& m) { return m && !m->isEmpty() && m->planeCount() == 1 && m->elmtInBytes() == sizeof(uint8_t); }

// The output plane: same geometry and stride as the input, reused when the caller's mat is large enough, possibly the input
// itself (samples/edges_canny/main.cxx:72 calls process(mat, &mat)); `keep` holds the input alive across a reallocation.
static COMPV_ERROR_CODE outputLike(const CompVMatPtr& image, CompVMatPtrPtr edges, CompVMatPtr& keep)
{
	keep = image;
	return CompVImage::newObj8u(edges, COMPV_SUBTYPE_PIXELS_Y, image->cols(), image->rows(), image->stride());
}

// Calls `run(buffer, capacity, &n)` until the host buffer holds every line, then converts to CompVHoughLine.
template <typename Run>
static COMPV_ERROR_CODE collectLines(compvhip_ctx* ctx, const char* what, std::vector<compvhip_line>& buf, CompVHoughLineVector& lines, Run run)
{
	lines.clear();
	if (buf.size() < 4096) buf.resize(4096);
	size_t n = 0;
	int rc = run(buf.data(), buf.size(), &n);
	if (rc == COMPVHIP_E_OUT_OF_BOUND) { buf.resize(n); rc = run(buf.data(), buf.size(), &n); }
	if (rc != COMPVHIP_OK) return report(ctx, what, rc);
	lines.reserve(n);
	for (size_t i = 0; i < n; ++i) lines.push_back(CompVHoughLine(buf[i].rho, buf[i].theta, static_cast<size_t>(buf[i].strength)));
	return COMPV_ERROR_CODE_S_OK;
}

// polar -> cartesian through the C ABI (compvhip_houghsht_to_cartesian / compvhip_houghkht_to_cartesian)
static COMPV_ERROR_CODE cartesian(int (*fn)(size_t, size_t, const compvhip_line*, size_t, float*), size_t W, size_t H, const CompVHoughLineVector& polar,
	CompVLineFloat32Vector& out)
{
	out.clear();
	if (!W || !H) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
	const size_t n = polar.size();
	if (!n) return COMPV_ERROR_CODE_S_OK;
	std::vector<compvhip_line> in(n);
	std::vector<float> xy(4 * n);
	for (size_t i = 0; i < n; ++i) { in[i].rho = polar[i].rho; in[i].theta = polar[i].theta; }
	const int rc = fn(W, H, in.data(), n, xy.data());
	if (rc != COMPVHIP_OK) return translate(rc);
	out.resize(n);
	for (size_t i = 0; i < n; ++i) {
		out[i].a.x = xy[4 * i]; out[i].a.y = xy[4 * i + 1]; out[i].b.x = xy[4 * i + 2]; out[i].b.y = xy[4 * i + 3];
		out[i].a.z = out[i].b.z = 1.f;
	}
	return COMPV_ERROR_CODE_S_OK;
}

// intrusive-refcounted construction shared by the four factories
template <typename T, typename Base, typename A, typename B, typename C>
static COMPV_ERROR_CODE make(CompVPtr<Base*>* out, A a, B b, C c)
{
	if (!out) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
	CompVPtr<T*> obj = new T(a, b, c);
	if (!obj) return COMPV_ERROR_CODE_E_OUT_OF_MEMORY;
	*out = *obj;
	return COMPV_ERROR_CODE_S_OK;
}

} // namespace hipbinding

using namespace hipbinding;

// ------------------------------------------------------------------------------------------------------------------
// Canny  (stands in for CompVEdgeDeteCanny)
// ------------------------------------------------------------------------------------------------------------------
class CompVEdgeDeteCannyHip : public CompVEdgeDete
{
public:
	CompVEdgeDeteCannyHip(float tLow, float tHigh, size_t kernSize) : CompVEdgeDete(COMPV_CANNY_ID), m_Set(defaults())
	{
		m_Set.cannyLow = tLow; m_Set.cannyHigh = tHigh; m_Set.kernelSize = (kernSize == 3) ? 3 : 5;
	}
	virtual ~CompVEdgeDeteCannyHip() { }
	COMPV_OBJECT_GET_ID(CompVEdgeDeteCannyHip);

	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		return applyOption(kCannyOptions, OPT_COUNT(kCannyOptions), m_Set, id, valuePtr, valueSize, true, this);
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& image, CompVMatPtrPtr edges, CompVMatPtrPtr directions = NULL) override
	{
		if (!isGray8(image) || !edges) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		if (m_Set.cannyLow >= m_Set.cannyHigh) return COMPV_ERROR_CODE_E_INVALID_STATE; // canny_dete.cxx:126
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.acquire(ctx));
		CompVMatPtr in;
		COMPV_CHECK_CODE_RETURN(outputLike(image, edges, in));
		const int type = (m_Set.cannyType == COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN) ? COMPVHIP_CANNY_THRESHOLD_PERCENT_OF_MEAN : COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT;
		const int rc = compvhip_canny_u8(ctx, in->ptr<const uint8_t>(), in->cols(), in->rows(), in->stride(), m_Set.cannyLow, m_Set.cannyHigh, m_Set.kernelSize, type,
			(*edges)->ptr<uint8_t>(), (*edges)->stride());
		if (directions) *directions = NULL; // the reference never allocates its direction map either (canny_dete.cxx:150-155)
		return report(ctx, "compvhip_canny_u8", rc);
	}

	static COMPV_ERROR_CODE newObj(CompVEdgeDetePtrPtr dete, float tLow, float tHigh, size_t kernSize) { return make<CompVEdgeDeteCannyHip>(dete, tLow, tHigh, kernSize); }

private:
	LazyCtx m_Ctx;
	Settings m_Set;
};

// ------------------------------------------------------------------------------------------------------------------
// Sobel / Scharr / Prewitt  (stand in for CompVCornerDeteEdgeBase; the factory arguments are ignored as in edge_dete.cxx:208-224)
// ------------------------------------------------------------------------------------------------------------------
class CompVEdgeDeteBaseHip : public CompVEdgeDete
{
public:
	CompVEdgeDeteBaseHip(int id, int op, int) : CompVEdgeDete(id), m_nOp(op) { }
	virtual ~CompVEdgeDeteBaseHip() { }
	COMPV_OBJECT_GET_ID(CompVEdgeDeteBaseHip);

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& image, CompVMatPtrPtr edges, CompVMatPtrPtr directions = NULL) override
	{
		if (!image || image->subType() != COMPV_SUBTYPE_PIXELS_Y || !edges) return COMPV_ERROR_CODE_E_INVALID_PARAMETER; // edge_dete.cxx:57
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.acquire(ctx));
		CompVMatPtr in;
		COMPV_CHECK_CODE_RETURN(outputLike(image, edges, in));
		const int rc = compvhip_edge_dete_u8(ctx, in->ptr<const uint8_t>(), in->cols(), in->rows(), in->stride(), m_nOp, (*edges)->ptr<uint8_t>(), (*edges)->stride());
		if (directions) *directions = NULL;
		return report(ctx, "compvhip_edge_dete_u8", rc);
	}

	template <int ID, int OP>
	static COMPV_ERROR_CODE newObjOf(CompVEdgeDetePtrPtr dete, float, float, size_t) { return make<CompVEdgeDeteBaseHip>(dete, ID, OP, 0); }

private:
	LazyCtx m_Ctx;
	int m_nOp;
};

// ------------------------------------------------------------------------------------------------------------------
// Hough SHT / KHT  (stand in for CompVHoughSht, CompVHoughKht)
// ------------------------------------------------------------------------------------------------------------------
class CompVHoughShtHip : public CompVHough
{
public:
	CompVHoughShtHip(float rho, float thetaDeg, size_t threshold) : CompVHough(COMPV_HOUGHSHT_ID), m_Set(defaults())
	{
		m_Set.rho = rho; m_Set.thetaDeg = thetaDeg; m_Set.threshold = static_cast<int>(threshold);
	}
	virtual ~CompVHoughShtHip() { }
	COMPV_OBJECT_GET_ID(CompVHoughShtHip);

	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		const float rhoBefore = m_Set.rho;
		const COMPV_ERROR_CODE err = applyOption(kShtOptions, OPT_COUNT(kShtOptions), m_Set, id, valuePtr, valueSize, false, this);
		if (COMPV_ERROR_CODE_IS_OK(err) && m_Set.rho != 1.f) { // SHT only exists for rho == 1 (houghsht.cxx:306-316): refuse and keep the old value
			m_Set.rho = rhoBefore;
			COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "SHT requires rho == 1; use KHT for fractional rho");
			return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		}
		return err;
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& edges, CompVHoughLineVector& lines, const CompVMatPtr& directions = NULL) override
	{
		if (!isGray8(edges)) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.acquire(ctx));
		const Settings& s = m_Set;
		return collectLines(ctx, "compvhip_houghsht_u8", m_Lines, lines, [&](compvhip_line* buf, size_t cap, size_t* n) {
			return compvhip_houghsht_u8(ctx, edges->ptr<const uint8_t>(), edges->cols(), edges->rows(), edges->stride(), s.rho, s.thetaDeg, s.threshold, s.maxLines, buf, cap, n, NULL, 0);
		});
	}

	virtual COMPV_ERROR_CODE toCartesian(const size_t imageWidth, const size_t imageHeight, const CompVHoughLineVector& polar, CompVLineFloat32Vector& cart) override
	{
		return cartesian(compvhip_houghsht_to_cartesian, imageWidth, imageHeight, polar, cart);
	}

	static COMPV_ERROR_CODE newObj(CompVHoughPtrPtr hough, float rho, float theta, size_t threshold)
	{
		if (rho != 1.f) return COMPV_ERROR_CODE_E_INVALID_PARAMETER; // covers rho <= 0 and rho > 1 as well
		return make<CompVHoughShtHip>(hough, rho, theta, threshold);
	}

private:
	LazyCtx m_Ctx;
	Settings m_Set;
	std::vector<compvhip_line> m_Lines;
};

class CompVHoughKhtHip : public CompVHough
{
public:
	CompVHoughKhtHip(float rho, float thetaDeg, size_t threshold)
		: CompVHough(COMPV_HOUGHKHT_ID), m_Set(defaults()), m_dMinDeviation(2.0), m_dMinHeight(0.002), m_dGS(1.0) // double defaults: houghkht.cxx:38-40
	{
		m_Set.rho = rho; m_Set.thetaDeg = thetaDeg; m_Set.threshold = static_cast<int>(threshold);
	}
	virtual ~CompVHoughKhtHip() { }
	COMPV_OBJECT_GET_ID(CompVHoughKhtHip);

	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		const COMPV_ERROR_CODE err = applyOption(kKhtOptions, OPT_COUNT(kKhtOptions), m_Set, id, valuePtr, valueSize, false, this);
		if (COMPV_ERROR_CODE_IS_OK(err)) { // the two float32 knobs are kept as doubles, like the reference's members (their defaults are double literals)
			if (id == COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION) m_dMinDeviation = static_cast<double>(m_Set.clusterMinDeviation);
			if (id == COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH) m_dMinHeight = static_cast<double>(m_Set.kernelMinHeight);
		}
		return err;
	}

	// COMPV_HOUGHKHT_GET_FLT64_GS (houghkht.cxx:194-206): the caller passes the address of a pointer to its own double
	virtual COMPV_ERROR_CODE get(int id, const void** valuePtrPtr, size_t valueSize) override
	{
		if (!valuePtrPtr || !valueSize) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		if (id != COMPV_HOUGHKHT_GET_FLT64_GS) return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED;
		if (valueSize != sizeof(compv_float64_t)) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		memcpy(const_cast<void*>(*valuePtrPtr), &m_dGS, sizeof(m_dGS));
		return COMPV_ERROR_CODE_S_OK;
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& edges, CompVHoughLineVector& lines, const CompVMatPtr& directions = NULL) override
	{
		if (!isGray8(edges)) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.acquire(ctx));
		const Settings& s = m_Set;
		return collectLines(ctx, "compvhip_houghkht_u8", m_Lines, lines, [&](compvhip_line* buf, size_t cap, size_t* n) {
			return compvhip_houghkht_u8(ctx, edges->ptr<const uint8_t>(), edges->cols(), edges->rows(), edges->stride(), s.rho, s.thetaDeg, s.threshold, s.maxLines,
				m_dMinDeviation, static_cast<size_t>(s.clusterMinSize), m_dMinHeight, buf, cap, n, &m_dGS);
		});
	}

	virtual COMPV_ERROR_CODE toCartesian(const size_t imageWidth, const size_t imageHeight, const CompVHoughLineVector& polar, CompVLineFloat32Vector& cart) override
	{
		return cartesian(compvhip_houghkht_to_cartesian, imageWidth, imageHeight, polar, cart);
	}

	static COMPV_ERROR_CODE newObj(CompVHoughPtrPtr hough, float rho, float theta, size_t threshold)
	{
		if (!(rho > 0.f) || rho > 1.f) return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
		return make<CompVHoughKhtHip>(hough, rho, theta, threshold);
	}

private:
	LazyCtx m_Ctx;
	Settings m_Set;
	double m_dMinDeviation, m_dMinHeight;
	double m_dGS;
	std::vector<compvhip_line> m_Lines;
};

// ------------------------------------------------------------------------------------------------------------------
// factory table (file-static: addFactory stores the POINTER, as core/compv_core.cxx:56-103 relies on for the built-ins)
// ------------------------------------------------------------------------------------------------------------------
static const CompVFeatureFactory kHipFactories[] = {
	{ COMPV_CANNY_ID, "Canny edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteCannyHip::newObj, nullptr, nullptr },
	{ COMPV_SOBEL_ID, "Sobel edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjOf<COMPV_SOBEL_ID, COMPVHIP_OP_SOBEL>, nullptr, nullptr },
	{ COMPV_SCHARR_ID, "Scharr edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjOf<COMPV_SCHARR_ID, COMPVHIP_OP_SCHARR>, nullptr, nullptr },
	{ COMPV_PREWITT_ID, "Prewitt edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjOf<COMPV_PREWITT_ID, COMPVHIP_OP_PREWITT>, nullptr, nullptr },
	{ COMPV_HOUGHSHT_ID, "Hough standard (HIP/gfx950)", nullptr, nullptr, nullptr, CompVHoughShtHip::newObj, nullptr },
	{ COMPV_HOUGHKHT_ID, "Hough kernel-based (HIP/gfx950)", nullptr, nullptr, nullptr, CompVHoughKhtHip::newObj, nullptr },
};

COMPV_NAMESPACE_END()

// Call once after CompVInit() / CompVCore::init(): swaps the CPU factories for the HIP ones.
extern "C" __attribute__((visibility("default"))) int compv_hip_plugin_register(void)
{
	using namespace compv;
	if (compvhip_device_count() <= 0) {
		COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "no HIP device: factories left untouched");
		return -1;
	}
	for (size_t i = 0; i < sizeof(kHipFactories) / sizeof(kHipFactories[0]); ++i) {
		if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&kHipFactories[i]))) return -2;
	}
	return 0;
}
