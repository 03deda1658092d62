// compv_hip_plugin.cxx -- the reference-side binding: CompV C++ classes that implement CompV's own abstract
// CompVEdgeDete / CompVHough interfaces on top of the C ABI in include/compv_hip.h, and the factory table that
// re-registers the ids COMPV_CANNY_ID / COMPV_SOBEL_ID / COMPV_SCHARR_ID / COMPV_PREWITT_ID / COMPV_HOUGHSHT_ID / COMPV_HOUGHKHT_ID.
//
// CompVFeature::addFactory() REPLACES an existing id (base/compv_features.cxx:30-40), so after
//     CompVInit(); compv_hip_plugin_register();
// every CompVEdgeDete::newObj(&p, COMPV_CANNY_ID, ...) / CompVHough::newObj(&h, COMPV_HOUGHSHT_ID, ...) in
// samples/edges_canny, samples/hough_lines and unittests/*.cxx returns the HIP implementation, unchanged call sites.
//
// This file is compiled against a CompV checkout (-I$COMPV_ROOT/base/include ...): it contains no CompV source, it
// only derives from the public abstract classes (base/include/compv/base/compv_features.h:207-227) exactly as
// core/features/edges/compv_core_feature_canny_dete.h:22-61 and core/features/hough/compv_core_feature_houghsht.h:57-81 do.
#include <compv/base/compv_base.h>
#include <compv/base/compv_features.h>
#include <compv/base/compv_debug.h>
#include <compv/base/image/compv_image.h>
#include <compv/base/math/compv_math.h>

#include "../include/compv_hip.h"

#include <climits>
#include <cmath>
#include <vector>

#define COMPV_THIS_CLASSNAME "CompVHipPlugin"

COMPV_NAMESPACE_BEGIN()

// COMPVHIP_E_* -> COMPV_ERROR_CODE (table in INTEGRATION.md)
static COMPV_ERROR_CODE toCompvError(int rc)
{
	switch (rc) {
	case COMPVHIP_OK: return COMPV_ERROR_CODE_S_OK;
	case COMPVHIP_E_NOT_IMPLEMENTED: return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED;
	case COMPVHIP_E_NOT_INITIALIZED: return COMPV_ERROR_CODE_E_NOT_INITIALIZED;
	case COMPVHIP_E_INVALID_STATE: return COMPV_ERROR_CODE_E_INVALID_STATE;
	case COMPVHIP_E_INVALID_PARAMETER: return COMPV_ERROR_CODE_E_INVALID_PARAMETER;
	case COMPVHIP_E_OUT_OF_MEMORY: return COMPV_ERROR_CODE_E_OUT_OF_MEMORY;
	case COMPVHIP_E_OUT_OF_BOUND: return COMPV_ERROR_CODE_E_OUT_OF_BOUND;
	default: return COMPV_ERROR_CODE_E_CUDA; // no HIP code exists in CompV; E_CUDA is the GPU-runtime class (compv_common.h:266)
	}
}

#define COMPVHIP_CHECK(ctx, call) do { const int rc__ = (call); if (rc__ != COMPVHIP_OK) { \
	COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "%s failed (%d): %s", #call, rc__, compvhip_last_error(ctx)); return toCompvError(rc__); } } while (0)

// One compvhip_ctx per detector object: a CompV detector instance owns its scratch and is not re-entrant
// (core/features/edges/compv_core_feature_canny_dete.cxx:133-147); the same holds here.
class CompVHipCtxHolder
{
public:
	CompVHipCtxHolder() : m_pCtx(NULL) { }
	~CompVHipCtxHolder() { if (m_pCtx) compvhip_ctx_destroy(m_pCtx); }
	COMPV_ERROR_CODE get(compvhip_ctx** ctx)
	{
		if (!m_pCtx) {
			const int rc = compvhip_ctx_create(&m_pCtx, -1);
			if (rc != COMPVHIP_OK) {
				COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "No usable HIP device (%d): the HIP plugin has no CPU fallback", rc);
				return toCompvError(rc);
			}
		}
		*ctx = m_pCtx;
		return COMPV_ERROR_CODE_S_OK;
	}
private:
	compvhip_ctx* m_pCtx;
};

// ------------------------------------------------------------------------------------------------------------------
// Canny  (replaces CompVEdgeDeteCanny)
// ------------------------------------------------------------------------------------------------------------------
class CompVEdgeDeteCannyHip : public CompVEdgeDete
{
protected:
	CompVEdgeDeteCannyHip(float tLow, float tHigh, size_t kernSize)
		: CompVEdgeDete(COMPV_CANNY_ID)
		, m_nKernelSize(kernSize == 3 ? 3 : 5)
		, m_fThresholdLow(tLow)
		, m_fThresholdHigh(tHigh)
		, m_nThresholdType(COMPV_CANNY_THRESHOLD_TYPE_COMPARE_TO_GRADIENT) { }
public:
	virtual ~CompVEdgeDeteCannyHip() { }
	COMPV_OBJECT_GET_ID(CompVEdgeDeteCannyHip);

	// same ids, value sizes and range checks as CompVEdgeDeteCanny::set (canny_dete.cxx:77-117)
	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		COMPV_CHECK_EXP_RETURN(!valuePtr || !valueSize, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		switch (id) {
		case COMPV_CANNY_SET_INT_THRESHOLD_TYPE: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int32_t), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			const int32_t t = *reinterpret_cast<const int32_t*>(valuePtr);
			COMPV_CHECK_EXP_RETURN(t != COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN && t != COMPV_CANNY_THRESHOLD_TYPE_COMPARE_TO_GRADIENT, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nThresholdType = t;
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_CANNY_SET_FLT32_THRESHOLD_LOW: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_fThresholdLow = *reinterpret_cast<const compv_float32_t*>(valuePtr);
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_CANNY_SET_FLT32_THRESHOLD_HIGH: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_fThresholdHigh = *reinterpret_cast<const compv_float32_t*>(valuePtr);
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_CANNY_SET_INT_KERNEL_SIZE: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int) || (*reinterpret_cast<const int*>(valuePtr) != 3 && *reinterpret_cast<const int*>(valuePtr) != 5), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nKernelSize = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		default:
			COMPV_CHECK_CODE_RETURN(CompVCaps::set(id, valuePtr, valueSize));
			return COMPV_ERROR_CODE_S_OK;
		}
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& image, CompVMatPtrPtr edges, CompVMatPtrPtr directions = NULL) override
	{
		COMPV_CHECK_EXP_RETURN(!image || image->planeCount() != 1 || image->elmtInBytes() != sizeof(uint8_t) || !edges, COMPV_ERROR_CODE_E_INVALID_PARAMETER, "Input image is null or invalid format");
		COMPV_CHECK_EXP_RETURN(m_fThresholdLow >= m_fThresholdHigh, COMPV_ERROR_CODE_E_INVALID_STATE, "Invalid state: m_fThresholdLow >= m_fThresholdHigh");
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.get(&ctx));
		const size_t W = image->cols(), H = image->rows(), S = image->stride();
		// like the reference, the output is created after the input has been consumed on the device side, has the
		// input's stride and may be the input itself (samples/edges_canny/main.cxx:72 calls process(mat, &mat))
		CompVMatPtr in = image; // keeps the input alive if *edges == image and newObj8u reallocates
		COMPV_CHECK_CODE_RETURN(CompVImage::newObj8u(edges, COMPV_SUBTYPE_PIXELS_Y, W, H, S));
		COMPVHIP_CHECK(ctx, compvhip_canny_u8(ctx, in->ptr<const uint8_t>(), W, H, S, m_fThresholdLow, m_fThresholdHigh,
			static_cast<int>(m_nKernelSize),
			m_nThresholdType == COMPV_CANNY_THRESHOLD_TYPE_PERCENT_OF_MEAN ? COMPVHIP_CANNY_THRESHOLD_PERCENT_OF_MEAN : COMPVHIP_CANNY_THRESHOLD_COMPARE_TO_GRADIENT,
			(*edges)->ptr<uint8_t>(), (*edges)->stride()));
		if (directions) {
			*directions = NULL; // the reference never allocates gradDir either (canny_dete.cxx:150-155)
		}
		return COMPV_ERROR_CODE_S_OK;
	}

	static COMPV_ERROR_CODE newObj(CompVEdgeDetePtrPtr dete, float tLow, float tHigh, size_t kernSize)
	{
		COMPV_CHECK_EXP_RETURN(!dete, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		CompVPtr<CompVEdgeDeteCannyHip*> dete_ = new CompVEdgeDeteCannyHip(tLow, tHigh, kernSize);
		COMPV_CHECK_EXP_RETURN(!dete_, COMPV_ERROR_CODE_E_OUT_OF_MEMORY);
		*dete = *dete_;
		return COMPV_ERROR_CODE_S_OK;
	}

private:
	CompVHipCtxHolder m_Ctx;
	size_t m_nKernelSize;
	float m_fThresholdLow;
	float m_fThresholdHigh;
	int m_nThresholdType;
};

// ------------------------------------------------------------------------------------------------------------------
// Sobel / Scharr / Prewitt  (replaces CompVCornerDeteEdgeBase)
// ------------------------------------------------------------------------------------------------------------------
class CompVEdgeDeteBaseHip : public CompVEdgeDete
{
protected:
	CompVEdgeDeteBaseHip(int id, int op) : CompVEdgeDete(id), m_nOp(op) { }
public:
	virtual ~CompVEdgeDeteBaseHip() { }
	COMPV_OBJECT_GET_ID(CompVEdgeDeteBaseHip);

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& image, CompVMatPtrPtr edges, CompVMatPtrPtr directions = NULL) override
	{
		// edge_dete.cxx:57
		COMPV_CHECK_EXP_RETURN(!image || image->subType() != COMPV_SUBTYPE_PIXELS_Y || !edges, COMPV_ERROR_CODE_E_INVALID_PARAMETER, "Input image is null or not in grayscale format");
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.get(&ctx));
		const size_t W = image->cols(), H = image->rows(), S = image->stride();
		CompVMatPtr in = image;
		COMPV_CHECK_CODE_RETURN(CompVImage::newObj8u(edges, COMPV_SUBTYPE_PIXELS_Y, W, H, S));
		COMPVHIP_CHECK(ctx, compvhip_edge_dete_u8(ctx, in->ptr<const uint8_t>(), W, H, S, m_nOp, (*edges)->ptr<uint8_t>(), (*edges)->stride()));
		if (directions) *directions = NULL;
		return COMPV_ERROR_CODE_S_OK;
	}

	// tLow / tHigh / kernSize are ignored exactly as in the reference (edge_dete.cxx:208-224)
	static COMPV_ERROR_CODE newObjSobel(CompVEdgeDetePtrPtr dete, float, float, size_t) { return newObj(dete, COMPV_SOBEL_ID, COMPVHIP_OP_SOBEL); }
	static COMPV_ERROR_CODE newObjScharr(CompVEdgeDetePtrPtr dete, float, float, size_t) { return newObj(dete, COMPV_SCHARR_ID, COMPVHIP_OP_SCHARR); }
	static COMPV_ERROR_CODE newObjPrewitt(CompVEdgeDetePtrPtr dete, float, float, size_t) { return newObj(dete, COMPV_PREWITT_ID, COMPVHIP_OP_PREWITT); }

private:
	static COMPV_ERROR_CODE newObj(CompVEdgeDetePtrPtr dete, int id, int op)
	{
		COMPV_CHECK_EXP_RETURN(!dete, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		CompVPtr<CompVEdgeDeteBaseHip*> dete_ = new CompVEdgeDeteBaseHip(id, op);
		COMPV_CHECK_EXP_RETURN(!dete_, COMPV_ERROR_CODE_E_OUT_OF_MEMORY);
		*dete = *dete_;
		return COMPV_ERROR_CODE_S_OK;
	}
	CompVHipCtxHolder m_Ctx;
	int m_nOp;
};

// ------------------------------------------------------------------------------------------------------------------
// Hough SHT  (replaces CompVHoughSht)
// ------------------------------------------------------------------------------------------------------------------
class CompVHoughShtHip : public CompVHough
{
protected:
	CompVHoughShtHip(float rho, float thetaDeg, size_t threshold)
		: CompVHough(COMPV_HOUGHSHT_ID), m_fRho(rho), m_fThetaDeg(thetaDeg), m_nThreshold(threshold), m_nMaxLines(INT_MAX) { }
public:
	virtual ~CompVHoughShtHip() { }
	COMPV_OBJECT_GET_ID(CompVHoughShtHip);

	// same ids / checks as CompVHoughSht::set (houghsht.cxx:60-94)
	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		COMPV_CHECK_EXP_RETURN(!valuePtr || !valueSize, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		switch (id) {
		case COMPV_HOUGH_SET_FLT32_RHO: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f || *reinterpret_cast<const compv_float32_t*>(valuePtr) > 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			COMPV_CHECK_EXP_RETURN(*reinterpret_cast<const compv_float32_t*>(valuePtr) != 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER, "SHT implementation requires rho equal to 1f. Please use KHT implementation for fractional rho values");
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_FLT32_THETA: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_fThetaDeg = *reinterpret_cast<const compv_float32_t*>(valuePtr);
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_INT_THRESHOLD: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int) || *reinterpret_cast<const int*>(valuePtr) <= 0, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nThreshold = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_INT_MAXLINES: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nMaxLines = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr) <= 0 ? INT_MAX : *reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		default: {
			COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "Set with id %d not implemented", id);
			return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED;
		}
		}
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& edges, CompVHoughLineVector& lines, const CompVMatPtr& directions = NULL) override
	{
		COMPV_CHECK_EXP_RETURN(!edges || edges->isEmpty() || edges->planeCount() != 1 || edges->elmtInBytes() != sizeof(uint8_t), COMPV_ERROR_CODE_E_INVALID_PARAMETER, "Edges null or not 8uc1");
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.get(&ctx));
		lines.clear();
		const int maxLines = (m_nMaxLines >= static_cast<size_t>(INT_MAX)) ? 0 : static_cast<int>(m_nMaxLines);
		size_t n = 0;
		if (m_Lines.size() < 4096) m_Lines.resize(4096);
		int rc = compvhip_houghsht_u8(ctx, edges->ptr<const uint8_t>(), edges->cols(), edges->rows(), edges->stride(), m_fRho, m_fThetaDeg,
			static_cast<int>(m_nThreshold), maxLines, m_Lines.data(), m_Lines.size(), &n, NULL, 0);
		if (rc == COMPVHIP_E_OUT_OF_BOUND) { // grow the host buffer once and ask again
			m_Lines.resize(n);
			rc = compvhip_houghsht_u8(ctx, edges->ptr<const uint8_t>(), edges->cols(), edges->rows(), edges->stride(), m_fRho, m_fThetaDeg,
				static_cast<int>(m_nThreshold), maxLines, m_Lines.data(), m_Lines.size(), &n, NULL, 0);
		}
		COMPVHIP_CHECK(ctx, rc);
		lines.reserve(n);
		for (size_t i = 0; i < n; ++i) {
			lines.push_back(CompVHoughLine(m_Lines[i].rho, m_Lines[i].theta, static_cast<size_t>(m_Lines[i].strength)));
		}
		return COMPV_ERROR_CODE_S_OK;
	}

	// polar -> cartesian: tiny, float32, stays on the host; same arithmetic as houghsht.cxx:566-605
	virtual COMPV_ERROR_CODE toCartesian(const size_t imageWidth, const size_t imageHeight, const CompVHoughLineVector& polar, CompVLineFloat32Vector& cartesian) override
	{
		COMPV_CHECK_EXP_RETURN(!imageWidth || !imageHeight, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		cartesian.clear();
		if (polar.empty()) return COMPV_ERROR_CODE_S_OK;
		cartesian.resize(polar.size());
		const compv_float32_t widthF = static_cast<compv_float32_t>(imageWidth);
		const compv_float32_t heightF = static_cast<compv_float32_t>(imageHeight);
		const compv_float32_t r = std::sqrt((widthF * widthF) + (heightF * heightF));
		for (size_t k = 0; k < polar.size(); ++k) {
			const compv_float32_t theta = polar[k].theta, rho = polar[k].rho;
			CompVLineFloat32& l = cartesian[k];
			if (theta == 0.f) { // perfect vertical line
				l.a.x = l.b.x = rho;
				l.a.y = r;
				l.b.y = -r;
			}
			else {
				const compv_float32_t a = std::cos(theta), b = (1.f / std::sin(theta));
				l.a.x = 0.f;
				l.a.y = (rho * b);
				l.b.x = widthF;
				l.b.y = ((rho - (l.b.x * a)) * b);
			}
			l.a.z = l.b.z = 1.f;
		}
		return COMPV_ERROR_CODE_S_OK;
	}

	static COMPV_ERROR_CODE newObj(CompVHoughPtrPtr hough, float rho, float theta, size_t threshold)
	{
		COMPV_CHECK_EXP_RETURN(!hough || rho <= 0 || rho > 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		COMPV_CHECK_EXP_RETURN(rho != 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER, "SHT implementation requires rho equal to 1f. Please use KHT implementation for fractional rho values");
		CompVPtr<CompVHoughShtHip*> hough_ = new CompVHoughShtHip(rho, theta, threshold);
		COMPV_CHECK_EXP_RETURN(!hough_, COMPV_ERROR_CODE_E_OUT_OF_MEMORY);
		*hough = *hough_;
		return COMPV_ERROR_CODE_S_OK;
	}

private:
	CompVHipCtxHolder m_Ctx;
	float m_fRho;
	float m_fThetaDeg;
	size_t m_nThreshold;
	size_t m_nMaxLines;
	std::vector<compvhip_line> m_Lines;
};

// ------------------------------------------------------------------------------------------------------------------
// Hough KHT  (replaces CompVHoughKht)
// ------------------------------------------------------------------------------------------------------------------
class CompVHoughKhtHip : public CompVHough
{
protected:
	CompVHoughKhtHip(float rho, float thetaDeg, size_t threshold)
		: CompVHough(COMPV_HOUGHKHT_ID), m_fRho(rho), m_fThetaDeg(thetaDeg), m_nThreshold(threshold), m_nMaxLines(INT_MAX)
		, m_dClusterMinDeviation(2.0), m_nClusterMinSize(10), m_dKernelMinHeight(0.002), m_dGS(1.0) { } // houghkht.cxx:38-40,117-129
public:
	virtual ~CompVHoughKhtHip() { }
	COMPV_OBJECT_GET_ID(CompVHoughKhtHip);

	// ids / checks of CompVHoughKht::set (houghkht.cxx:140-192)
	virtual COMPV_ERROR_CODE set(int id, const void* valuePtr, size_t valueSize) override
	{
		COMPV_CHECK_EXP_RETURN(!valuePtr || !valueSize, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		switch (id) {
		case COMPV_HOUGH_SET_FLT32_RHO: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f || *reinterpret_cast<const compv_float32_t*>(valuePtr) > 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_fRho = *reinterpret_cast<const compv_float32_t*>(valuePtr);
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_FLT32_THETA: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_fThetaDeg = *reinterpret_cast<const compv_float32_t*>(valuePtr);
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_INT_THRESHOLD: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int) || *reinterpret_cast<const int*>(valuePtr) <= 0, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nThreshold = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGH_SET_INT_MAXLINES: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nMaxLines = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr) <= 0 ? INT_MAX : *reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGHKHT_SET_FLT32_CLUSTER_MIN_DEVIATION: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_dClusterMinDeviation = static_cast<double>(*reinterpret_cast<const compv_float32_t*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGHKHT_SET_INT_CLUSTER_MIN_SIZE: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(int) || *reinterpret_cast<const int*>(valuePtr) <= 0, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_nClusterMinSize = static_cast<size_t>(*reinterpret_cast<const int*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGHKHT_SET_FLT32_KERNEL_MIN_HEIGTH: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float32_t) || *reinterpret_cast<const compv_float32_t*>(valuePtr) <= 0.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			m_dKernelMinHeight = static_cast<double>(*reinterpret_cast<const compv_float32_t*>(valuePtr));
			return COMPV_ERROR_CODE_S_OK;
		}
		case COMPV_HOUGHKHT_SET_BOOL_OVERRIDE_INPUT_EDGES: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(bool), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			return COMPV_ERROR_CODE_S_OK; // the HIP path always works on its own copy: nothing to override
		}
		default: {
			COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "Set with id %d not implemented", id);
			return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED;
		}
		}
	}

	// COMPV_HOUGHKHT_GET_FLT64_GS (houghkht.cxx:194-206)
	virtual COMPV_ERROR_CODE get(int id, const void** valuePtrPtr, size_t valueSize) override
	{
		COMPV_CHECK_EXP_RETURN(!valuePtrPtr || !valueSize, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		switch (id) {
		case COMPV_HOUGHKHT_GET_FLT64_GS: {
			COMPV_CHECK_EXP_RETURN(valueSize != sizeof(compv_float64_t), COMPV_ERROR_CODE_E_INVALID_PARAMETER);
			*reinterpret_cast<compv_float64_t*>(const_cast<void*>(*valuePtrPtr)) = m_dGS;
			return COMPV_ERROR_CODE_S_OK;
		}
		default:
			return COMPV_ERROR_CODE_E_NOT_IMPLEMENTED;
		}
	}

	virtual COMPV_ERROR_CODE process(const CompVMatPtr& edges, CompVHoughLineVector& lines, const CompVMatPtr& directions = NULL) override
	{
		COMPV_CHECK_EXP_RETURN(!edges || edges->isEmpty(), COMPV_ERROR_CODE_E_INVALID_PARAMETER, "Edges null or empty");
		COMPV_CHECK_EXP_RETURN(edges->elmtInBytes() != sizeof(uint8_t) || edges->planeCount() != 1, COMPV_ERROR_CODE_E_INVALID_PARAMETER, "Edges must be 8U_1D (e.g. grayscale image)");
		compvhip_ctx* ctx = NULL;
		COMPV_CHECK_CODE_RETURN(m_Ctx.get(&ctx));
		lines.clear();
		const int maxLines = (m_nMaxLines >= static_cast<size_t>(INT_MAX)) ? 0 : static_cast<int>(m_nMaxLines);
		size_t n = 0;
		if (m_Lines.size() < 4096) m_Lines.resize(4096);
		int rc = COMPVHIP_OK;
		for (int attempt = 0; attempt < 2; ++attempt) {
			rc = compvhip_houghkht_u8(ctx, edges->ptr<const uint8_t>(), edges->cols(), edges->rows(), edges->stride(), m_fRho, m_fThetaDeg,
				static_cast<int>(m_nThreshold), maxLines, m_dClusterMinDeviation, m_nClusterMinSize, m_dKernelMinHeight, m_Lines.data(), m_Lines.size(), &n, &m_dGS);
			if (rc != COMPVHIP_E_OUT_OF_BOUND) break;
			m_Lines.resize(n);
		}
		COMPVHIP_CHECK(ctx, rc);
		lines.reserve(n);
		for (size_t i = 0; i < n; ++i) lines.push_back(CompVHoughLine(m_Lines[i].rho, m_Lines[i].theta, static_cast<size_t>(m_Lines[i].strength)));
		return COMPV_ERROR_CODE_S_OK;
	}

	// rho is measured from the image centre: same arithmetic as houghkht.cxx:1249-1280
	virtual COMPV_ERROR_CODE toCartesian(const size_t imageWidth, const size_t imageHeight, const CompVHoughLineVector& polar, CompVLineFloat32Vector& cartesian) override
	{
		COMPV_CHECK_EXP_RETURN(!imageWidth || !imageHeight, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		cartesian.clear();
		if (polar.empty()) return COMPV_ERROR_CODE_S_OK;
		cartesian.resize(polar.size());
		const compv_float32_t widthF = static_cast<compv_float32_t>(imageWidth);
		const compv_float32_t heightF = static_cast<compv_float32_t>(imageHeight);
		const compv_float32_t r = std::sqrt((widthF * widthF) + (heightF * heightF));
		const compv_float32_t half_widthF = widthF * 0.5f, half_heightF = heightF * 0.5f;
		for (size_t k = 0; k < polar.size(); ++k) {
			const compv_float32_t rho = polar[k].rho, theta = polar[k].theta;
			CompVLineFloat32& l = cartesian[k];
			if (theta == 0.f) {
				l.a.x = l.b.x = (rho + half_widthF);
				l.a.y = r;
				l.b.y = -r;
			}
			else {
				const compv_float32_t a = (std::cos(theta) * half_widthF), b = (1.f / std::sin(theta));
				l.a.x = 0;
				l.a.y = ((rho + a) * b) + half_heightF;
				l.b.x = widthF;
				l.b.y = ((rho - a) * b) + half_heightF;
			}
			l.a.z = l.b.z = 1.f;
		}
		return COMPV_ERROR_CODE_S_OK;
	}

	static COMPV_ERROR_CODE newObj(CompVHoughPtrPtr hough, float rho, float theta, size_t threshold)
	{
		COMPV_CHECK_EXP_RETURN(!hough || rho <= 0 || rho > 1.f, COMPV_ERROR_CODE_E_INVALID_PARAMETER);
		CompVPtr<CompVHoughKhtHip*> hough_ = new CompVHoughKhtHip(rho, theta, threshold);
		COMPV_CHECK_EXP_RETURN(!hough_, COMPV_ERROR_CODE_E_OUT_OF_MEMORY);
		*hough = *hough_;
		return COMPV_ERROR_CODE_S_OK;
	}

private:
	CompVHipCtxHolder m_Ctx;
	float m_fRho;
	float m_fThetaDeg;
	size_t m_nThreshold;
	size_t m_nMaxLines;
	double m_dClusterMinDeviation;
	size_t m_nClusterMinSize;
	double m_dKernelMinHeight;
	double m_dGS;
	std::vector<compvhip_line> m_Lines;
};

// ------------------------------------------------------------------------------------------------------------------
// factory table (file-static: addFactory stores the POINTER, core/compv_core.cxx:56-103 does the same)
// ------------------------------------------------------------------------------------------------------------------
static const CompVFeatureFactory cannyHipFactory = { COMPV_CANNY_ID, "Canny edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteCannyHip::newObj, nullptr, nullptr };
static const CompVFeatureFactory sobelHipFactory = { COMPV_SOBEL_ID, "Sobel edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjSobel, nullptr, nullptr };
static const CompVFeatureFactory scharrHipFactory = { COMPV_SCHARR_ID, "Scharr edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjScharr, nullptr, nullptr };
static const CompVFeatureFactory prewittHipFactory = { COMPV_PREWITT_ID, "Prewitt edge detector (HIP/gfx950)", nullptr, nullptr, CompVEdgeDeteBaseHip::newObjPrewitt, nullptr, nullptr };
static const CompVFeatureFactory houghShtHipFactory = { COMPV_HOUGHSHT_ID, "Hough standard (HIP/gfx950)", nullptr, nullptr, nullptr, CompVHoughShtHip::newObj, nullptr };
static const CompVFeatureFactory houghKhtHipFactory = { COMPV_HOUGHKHT_ID, "Hough kernel-based (HIP/gfx950)", nullptr, nullptr, nullptr, CompVHoughKhtHip::newObj, nullptr };

COMPV_NAMESPACE_END()

// Call once after CompVInit() / CompVCore::init(): swaps the CPU factories for the HIP ones.
extern "C" __attribute__((visibility("default"))) int compv_hip_plugin_register(void)
{
	using namespace compv;
	if (compvhip_device_count() <= 0) {
		COMPV_DEBUG_ERROR_EX(COMPV_THIS_CLASSNAME, "no HIP device: factories left untouched");
		return -1;
	}
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&cannyHipFactory))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&sobelHipFactory))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&scharrHipFactory))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&prewittHipFactory))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&houghShtHipFactory))) return -2;
	if (COMPV_ERROR_CODE_IS_NOK(CompVFeature::addFactory(&houghKhtHipFactory))) return -2;
	return 0;
}
