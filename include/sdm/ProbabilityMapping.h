/*
 * sdm/ProbabilityMapping.h -- the reference's ProbabilityMapping class surface over the MI355X
 * engine (C ABI: include/sdm_c.h).
 *
 * Mirrors /root/reference/include/Modeler/ProbabilityMapping.h:61-105 ("PM.h"): same class name,
 * same public method names, same argument order and meaning, same void returns (errors are printed
 * to std::cerr, the convention of the surrounding Modeler code,
 * src/Modeler/SFMTranscriptInterface_ORBSLAM.cpp:213-215).  The only substitutions are the types
 * the container cannot provide (no OpenCV, no ORB_SLAM2 headers):
 *
 *     cv::Mat (CV_32FC1 / CV_8UC1)   ->  sdm::Mat<float> / sdm::Mat<uint8_t>  (rows, cols, at(y,x), clone())
 *     ORB_SLAM2::KeyFrame            ->  sdm::KeyFrame  (exactly the members PM.cc touches)
 *     ORB_SLAM2::Map                 ->  sdm::Map       (GetAllKeyFrames(), include/Map.h:58)
 *
 * INTEGRATION.md shows the adapter a maintainer of the fork writes to fill sdm::KeyFrame from the
 * real cv::Mat / ORB_SLAM2::KeyFrame.  All arithmetic runs on the GPU; this header and
 * host/ProbabilityMapping.cc only move data and keep the per-keyframe flags.
 */
#ifndef SDM_PROBABILITY_MAPPING_H
#define SDM_PROBABILITY_MAPPING_H

#include <cstdint>
#include <iosfwd>
#include <map>
#include <mutex>
#include <string>
#include <vector>

/* PM.h:38-49.  covisN is a runtime value here (sdm::Options::covisN, default 7). */
#define SDM_COVISN_DEFAULT 7

struct sdm_ctx;

namespace sdm {

/* Minimal dense 2-D array with cv::Mat-like value semantics: copies share nothing. */
template <typename T>
class Mat {
public:
    int rows = 0, cols = 0;
    std::vector<T> data;
    Mat() {}
    Mat(int r, int c, T v = T()) : rows(r), cols(c), data((size_t)r * (size_t)c, v) {}
    T& at(int y, int x) { return data[(size_t)y * cols + x]; }
    const T& at(int y, int x) const { return data[(size_t)y * cols + x]; }
    Mat clone() const { return *this; }
    bool empty() const { return data.empty(); }
    T* ptr() { return data.data(); }
    const T* ptr() const { return data.data(); }
};

/* The KeyFrame contract PM.cc assumes (use sites in PM.cc given per member). */
class KeyFrame {
public:
    long unsigned int mnId = 0;
    Mat<uint8_t> im_;             /* PM.cc:114,202,433   CV_8UC1 gray                      */
    Mat<float> GradImg;           /* PM.cc:201,411,434   gradient magnitude                */
    Mat<float> GradTheta;         /* PM.cc:214,415,427   gradient direction, deg [0,360)   */
    float I_stddev = 0.f;         /* PM.cc:457                                              */
    Mat<float> depth_map_;        /* PM.cc:225,237,345   inverse depth, zero-initialised   */
    Mat<float> depth_sigma_;      /* PM.cc:226,237                                          */
    Mat<float> SemiDensePointSets_; /* PM.cc:117-119,346-363  H x 3W                        */
    bool semidense_flag_ = false;   /* PM.cc:141,244 */
    bool interKF_depth_flag_ = false; /* PM.cc:265,306 */
    bool poseChanged = false;         /* PM.cc:329-331 */
    float fx = 0, fy = 0, cx = 0, cy = 0; /* include/KeyFrame.h:162 */
    float Tcw[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}; /* [R|t] row-major, src/KeyFrame.cc:70-121 */
    bool bad = false, mapped = true;
    std::vector<KeyFrame*> covisible;  /* GetVectorCovisibleKeyFrames(): descending weight, KeyFrame.cc:168-172 */
    std::vector<float> point_depths;   /* GetAllPointDepths(), PM.cc:371 */
    std::vector<int> map_point_ids;    /* GetMapPointMatches(): id per keypoint, -1 = none, PM.cc:468-469 */
    std::vector<float> keypoint_angles; /* GetKeyPointsUn()[i].angle, deg, <0 = none, PM.cc:475-476 */

    bool isBad() const { return bad; }
    bool Mapped() const { return mapped; }
    const Mat<uint8_t>& GetImage() const { return im_; }
    const std::vector<KeyFrame*>& GetVectorCovisibleKeyFrames() const { return covisible; }
    const std::vector<float>& GetAllPointDepths() const { return point_depths; }
    /* fills GradImg / GradTheta / I_stddev / zeroed maps from im_ ON THE GPU (the pre-processing the
     * reference leaves to KeyFrame); called lazily by ProbabilityMapping when GradImg is empty. */
};

class Map {
public:
    std::vector<KeyFrame*> keyframes;
    /* include/Map.h:58 / src/Map.cc:81-85: a copy taken under the map's mutex, so that a mapping thread
     * (ProbabilityMapping::Run) can poll while another thread inserts keyframes with AddKeyFrame */
    std::vector<KeyFrame*> GetAllKeyFrames() const
    {
        std::lock_guard<std::mutex> lock(mutex_);
        return keyframes;
    }
    void AddKeyFrame(KeyFrame* kf) /* src/Map.cc:38-44 */
    {
        std::lock_guard<std::mutex> lock(mutex_);
        keyframes.push_back(kf);
    }
    void EraseKeyFrame(KeyFrame* kf) /* src/Map.cc:55-66: the pointer leaves the map; the object is the caller's */
    {
        std::lock_guard<std::mutex> lock(mutex_);
        for (size_t i = 0; i < keyframes.size(); i++)
            if (keyframes[i] == kf) {
                keyframes.erase(keyframes.begin() + i);
                break;
            }
    }

private:
    mutable std::mutex mutex_;
};

/* What one rank does in a sharded pass (ProbabilityMapping::PlanBlock): keyframes are indices into the map's list. */
struct BlockPlan {
    std::vector<int> refs;                /* own keyframes reconstructed in this pass (good, not yet semidense, covisN neighbours) */
    std::vector<std::vector<int> > nbrs;  /* their neighbours, PM.cc:151-160 order */
    std::vector<int> check;               /* own keyframes checked in this pass: every neighbour has a map by then (PM.cc:292-298) */
    std::vector<std::vector<int> > check_nbrs;
    std::vector<char> needed;             /* [all]: keyframes this rank must hold = own block + the refs' and checks' neighbours */
    std::vector<char> boundary;           /* [all]: own keyframes some other rank's check reads: reconstruct first */
    std::vector<char> recon_all, check_all; /* [all]: reconstructed / checked in this pass by WHICHEVER rank owns it */
    std::vector<int> send_peer, send_kf;  /* maps that leave: (rank, keyframe), ascending per peer */
    std::vector<int> recv_peer, recv_kf;  /* maps that arrive; the k-th send to a peer is that peer's k-th receive */
};

struct Options {
    int device = 0;
    int covisN = SDM_COVISN_DEFAULT; /* PM.h:38 */
    int max_keyframes = 64;          /* device slots; least-recently-used keyframes are evicted */
    std::string obj_path = "semi_pointcloud.obj"; /* written when Run() ends, PM.cc:100 */
    unsigned poll_us = 5000;         /* Run()'s usleep, PM.cc:87 */
    bool exchange_compact = true;    /* sharded passes: maps cross ranks as their active-list entries (sdm_exchange_compact) */
};

}  // namespace sdm

class ProbabilityMapping {
public:
    struct depthHo { /* PM.h:64-70 (Pw is never written by PM.cc) */
        depthHo() : depth(0.0f), sigma(0.0f), supported(false) { Pw[0] = Pw[1] = Pw[2] = 0.0f; }
        float depth;
        float sigma;
        bool supported;
        float Pw[3];
    };

    explicit ProbabilityMapping(sdm::Map* pMap, const sdm::Options& opt = sdm::Options()); /* PM.h:72 */
    ~ProbabilityMapping();

    /* PM.cc:65-135: the mapping thread's loop -- poll the map, reconstruct every keyframe that is ready
     * (SemiDenseLoop), re-project after pose changes (UpdateAllSemiDensePointSet), sleep 5 ms; on RequestFinish
     * write semi_pointcloud.obj and return.  The context is single-caller: while Run() is active no other thread
     * may call into this object except RequestFinish / isFinished. */
    void Run();
    void SemiDenseLoop();            /* PM.cc:137-315 over every keyframe of the map (the form Run() calls at :76) */
    void RequestFinish();
    bool isFinished();
    long Passes();                   /* completed iterations of Run()'s loop (for callers that wait for the map to drain) */
    void SemiDenseRecon(sdm::KeyFrame* kf);                                               /* PM.h:75 */
    void StereoSearchConstraints(sdm::KeyFrame* kf, float* min_depth, float* max_depth);  /* PM.h:77 */
    void EpipolarSearch(sdm::KeyFrame* kf1, sdm::KeyFrame* kf2, const int x, const int y, float pixel,
                        float min_depth, float max_depth, depthHo* dh, const float F12[9], float& best_u,
                        float& best_v, float th_pi, float rot);                            /* PM.h:79 */
    void GetSearchRange(float& umin, float& umax, int px, int py, float mind, float maxd, sdm::KeyFrame* kf,
                        sdm::KeyFrame* kf2);                                               /* PM.h:80 */
    void InverseDepthHypothesisFusion(const std::vector<depthHo>& h, depthHo& dist);      /* PM.h:83 */
    void IntraKeyFrameDepthChecking(sdm::Mat<float>& depth_map, sdm::Mat<float>& depth_sigma,
                                    const sdm::Mat<float> gradimg);                        /* PM.h:85 */
    void IntraKeyFrameDepthGrowing(sdm::Mat<float>& depth_map, sdm::Mat<float>& depth_sigma,
                                   const sdm::Mat<float> gradimg);                         /* PM.h:86 */
    void UpdateSemiDensePointSet(sdm::KeyFrame* kf);                                      /* PM.h:88 */
    void UpdateAllSemiDensePointSet();                                                    /* PM.h:89 */
    void InterKeyFrameDepthChecking(sdm::KeyFrame* currentKf, std::vector<sdm::KeyFrame*> neighbors); /* PM.h:91 */

    /* PM.cc:972-986 (private in the reference; public here so tests can reach it) */
    void ComputeFundamental(sdm::KeyFrame* pKF1, sdm::KeyFrame* pKF2, float F12[9]);
    /* PM.cc:100-132: "v x y z" lines for sigma <= 0.01 and rho > 1e-6; returns the vertex count */
    long SavePointCloudObj(const char* path);
    /* The step after the path: the keyframe's semi-dense points as a CARV transcript entry,
     * in the exact text form of SFMTranscriptInterface_ORBSLAM::addKeyFrameInsertionWithLinesEntry
     * (src/Modeler/SFMTranscriptInterface_ORBSLAM.cpp:319-374):
     *     new cam: [x; y; z] {
     *     new point: [x; y; z], <camIndex>, <camIndexOriginal>
     *     }
     * points = pixels with sigma <= 0.01 and rho > 1e-6 (the obj writer's filter, PM.cc:120-121), raster
     * order; numbers through operator<<(double) like the reference.  Returns the number of points. */
    long AppendTranscriptEntry(sdm::KeyFrame* kf, int camIndex, int camIndexOriginal, std::ostream& out,
                               double max_sigma = 0.01);
    bool ok() const { return ctx_ != nullptr; }

    /* ---- device-slot cache contract (keyframes are cached by address) --------------------------------------
     * Forget: the integrator is about to delete (or has culled) the keyframe -- drops its device slot so that a
     * new KeyFrame allocated at the same address starts clean.  Call it where the fork erases keyframes
     * (src/KeyFrame.cc:449-497 SetBadFlag / Map::EraseKeyFrame, src/Map.cc:55-66).
     * InvalidateDepth: the integrator edited kf->depth_map_ / depth_sigma_ on the host (a sigma-threshold pass,
     * a manual reset): the next call that needs the map uploads the host copy again. */
    void Forget(sdm::KeyFrame* kf);
    void InvalidateDepth(sdm::KeyFrame* kf);

    /* ---- multi-GPU: one process per GPU, keyframes sharded in contiguous blocks -------------
     * InitSharding builds the engine's RCCL communicator from a 128-byte unique id (sdm_comm_unique_id on rank 0,
     * handed to the other ranks through any channel: a file, MPI, a socket).  world == 1 needs no id.
     * SemiDenseReconBlock runs the whole path for this rank's block all[first, first+count) of the map's
     * keyframes (`all` and the covisibility lists identical on every rank; images needed only for the block and
     * its covisible neighbours): SemiDenseRecon of the keyframes other ranks read -> their {rho,sigma} maps
     * leave over xGMI while the remaining keyframes are reconstructed -> InterKeyFrameDepthChecking against the
     * finished maps of all neighbours (snapshot order: every keyframe is checked against the neighbours' maps as
     * SemiDenseRecon left them, the order that shards) -> UpdateSemiDensePointSet.  Collective:
     * every rank calls it once per pass with its own block. */
    bool InitSharding(const unsigned char* comm_id, int world, int rank);
    /* the plan of a sharded pass, derived identically on every rank from the replicated covisibility lists (host only,
     * no device work: unit-tested on CPU against shard.plan, tests/test_adapter.py) */
    static bool PlanBlock(const std::vector<sdm::KeyFrame*>& all, int first, int count, int world, int rank, int covisN,
                          sdm::BlockPlan* out);
    void SemiDenseReconBlock(const std::vector<sdm::KeyFrame*>& all, int first, int count);
    /* diagnostic: would the device maps of these (resident) keyframes be accepted as SOURCES of the compact exchange --
     * pipeline maps, zero outside their pixel lists?  What SemiDenseReconBlock asks about the maps it will send before it
     * agrees on the pass's wire format (a "no" on any rank makes every rank move whole maps). */
    bool CompactSourcesReady(const std::vector<sdm::KeyFrame*>& kfs);

private:
    int SlotOf(sdm::KeyFrame* kf);   /* uploads the keyframe on first use */
    bool Ensure(int W, int H);
    void PushDepth(sdm::KeyFrame* kf, int slot);
    std::vector<sdm::KeyFrame*> PickNeighbours(sdm::KeyFrame* kf);  /* PM.cc:151-160 */
    static std::vector<sdm::KeyFrame*> PickNeighboursN(sdm::KeyFrame* kf, int covisN);

    bool CheckFinish();
    std::mutex mutex_finish_;
    bool finish_requested_ = false, finished_ = true;
    long passes_ = 0;
    sdm::Map* mpMap;
    sdm::Options opt_;
    sdm_ctx* ctx_ = nullptr;
    int W_ = 0, H_ = 0;
    std::map<sdm::KeyFrame*, int> slots_;
    std::map<sdm::KeyFrame*, int> depth_on_device_;  /* 1 = the slot's depth map equals kf->depth_map_ */
    std::map<sdm::KeyFrame*, float> map_lambdaG_;    /* lambdaG the keyframe's host map was reconstructed under: its support is
                                                        that keyframe's pixel list for THAT threshold (PM.cc:201) */
    float CurrentLambdaG() const;
    std::vector<sdm::KeyFrame*> slot_owner_;
    std::vector<unsigned long> slot_use_;
    unsigned long tick_ = 0;
};

#endif
