// host/ProbabilityMapping.cc -- the reference's ProbabilityMapping class surface
// (include/sdm/ProbabilityMapping.h) forwarding to the C ABI of the MI355X engine (include/sdm_c.h).
//
// Control flow follows /root/reference/src/Modeler/ProbabilityMapping.cc ("PM.cc") function by
// function; every numeric step happens on the GPU.  This file only: picks neighbours (PM.cc:151-160),
// derives the per-neighbour in-plane rotation and the depth prior from ORB data (host helpers of the C
// ABI), keeps keyframes resident in device slots, and mirrors results into the KeyFrame members the
// reference mutates (depth_map_, depth_sigma_, SemiDensePointSets_, the flags).
#include "sdm/ProbabilityMapping.h"

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>

#include <unistd.h>

#include "sdm_c.h"

namespace {
void report(const char* where)
{
    // same convention as the Modeler code around the seam: print and carry on
    std::cerr << "ProbabilityMapping::" << where << ": " << sdm_last_error() << std::endl;
}
}  // namespace

ProbabilityMapping::ProbabilityMapping(sdm::Map* pMap, const sdm::Options& opt) : mpMap(pMap), opt_(opt) {}

ProbabilityMapping::~ProbabilityMapping()
{
    if (ctx_) sdm_destroy(ctx_);
}

// PM.cc:65-135
void ProbabilityMapping::Run()
{
    {
        std::lock_guard<std::mutex> lock(mutex_finish_);
        finished_ = false;
    }
    while (1) {
        if (CheckFinish()) break;      // PM.cc:69
        SemiDenseLoop();               // PM.cc:76
        UpdateAllSemiDensePointSet();  // PM.cc:79-80: make point positions follow the keyframe poses
        {
            std::lock_guard<std::mutex> lock(mutex_finish_);
            passes_++;
        }
        usleep(opt_.poll_us);          // PM.cc:87
    }
    if (!opt_.obj_path.empty()) SavePointCloudObj(opt_.obj_path.c_str());  // PM.cc:100-132
    std::lock_guard<std::mutex> lock(mutex_finish_);
    finished_ = true;  // PM.cc:134
}

// the whole-map form of the loop (PM.cc:137-315 before its half-finished refactor to a per-keyframe call): every
// keyframe that is good, mapped and not reconstructed yet, in map order; SemiDenseRecon also runs the inter-keyframe
// phase (PM.cc:262-315) for every keyframe that became ready
void ProbabilityMapping::SemiDenseLoop()
{
    if (!mpMap) return;
    std::vector<sdm::KeyFrame*> vpKFs = mpMap->GetAllKeyFrames();
    for (size_t i = 0; i < vpKFs.size(); i++) {
        sdm::KeyFrame* kf = vpKFs[i];
        if (kf->isBad() || kf->semidense_flag_ || !kf->Mapped()) continue;
        SemiDenseRecon(kf);
    }
}

void ProbabilityMapping::RequestFinish()
{
    std::lock_guard<std::mutex> lock(mutex_finish_);
    finish_requested_ = true;
}

bool ProbabilityMapping::CheckFinish()
{
    std::lock_guard<std::mutex> lock(mutex_finish_);
    return finish_requested_;
}

long ProbabilityMapping::Passes()
{
    std::lock_guard<std::mutex> lock(mutex_finish_);
    return passes_;
}

bool ProbabilityMapping::isFinished()
{
    std::lock_guard<std::mutex> lock(mutex_finish_);
    return finished_;
}

bool ProbabilityMapping::Ensure(int W, int H)
{
    if (ctx_) {
        if (W != W_ || H != H_) {
            std::cerr << "ProbabilityMapping: image size " << W << "x" << H << " differs from the context's " << W_
                      << "x" << H_ << std::endl;
            return false;
        }
        return true;
    }
    sdm_config cfg;
    sdm_default_config(&cfg);
    cfg.device = opt_.device;
    cfg.W = W;
    cfg.H = H;
    cfg.max_keyframes = std::max(opt_.max_keyframes, opt_.covisN + 1);
    cfg.max_neighbours = std::max(1, std::min(opt_.covisN, SDM_MAX_NEIGHBOURS));
    cfg.with_pointset = 1;
    if (sdm_create(&ctx_, &cfg) != SDM_OK) {
        report("ProbabilityMapping");
        ctx_ = nullptr;
        return false;
    }
    W_ = W;
    H_ = H;
    slot_owner_.assign(cfg.max_keyframes, nullptr);
    slot_use_.assign(cfg.max_keyframes, 0);
    return true;
}

// Device slot of a keyframe; uploads its inputs on first use (least-recently-used eviction).
int ProbabilityMapping::SlotOf(sdm::KeyFrame* kf)
{
    if (!kf || kf->im_.empty()) return -1;
    if (!Ensure(kf->im_.cols, kf->im_.rows)) return -1;
    const float K[4] = {kf->fx, kf->fy, kf->cx, kf->cy};
    auto it = slots_.find(kf);
    if (it != slots_.end()) {
        slot_use_[it->second] = ++tick_;
        if (sdm_set_pose(ctx_, it->second, kf->Tcw) != SDM_OK) report("SlotOf");  // poses move under BA
        return it->second;
    }
    int slot = 0;
    for (size_t s = 0; s < slot_owner_.size(); s++) {
        if (!slot_owner_[s]) {
            slot = (int)s;
            break;
        }
        if (slot_use_[s] < slot_use_[slot]) slot = (int)s;
    }
    if (slot_owner_[slot]) {
        slots_.erase(slot_owner_[slot]);
        depth_on_device_.erase(slot_owner_[slot]);
    }
    int rc;
    if (kf->GradImg.empty() || kf->GradTheta.empty()) {
        // the pre-processing the reference leaves to KeyFrame (GradImg/GradTheta/I_stddev), on the GPU
        rc = sdm_upload_image(ctx_, slot, kf->im_.ptr(), K, kf->Tcw);
        if (rc == SDM_OK) {
            kf->GradImg = sdm::Mat<float>(H_, W_);
            kf->GradTheta = sdm::Mat<float>(H_, W_);
            rc = sdm_download_inputs(ctx_, slot, nullptr, kf->GradImg.ptr(), kf->GradTheta.ptr(), &kf->I_stddev);
        }
    } else {
        rc = sdm_upload_keyframe(ctx_, slot, kf->im_.ptr(), kf->GradImg.ptr(), kf->GradTheta.ptr(), kf->I_stddev, K,
                                 kf->Tcw);
    }
    if (rc != SDM_OK) {
        report("SlotOf");
        return -1;
    }
    if (kf->depth_map_.empty()) kf->depth_map_ = sdm::Mat<float>(H_, W_, 0.f);
    if (kf->depth_sigma_.empty()) kf->depth_sigma_ = sdm::Mat<float>(H_, W_, 0.f);
    if (kf->SemiDensePointSets_.empty()) kf->SemiDensePointSets_ = sdm::Mat<float>(H_, 3 * W_, 0.f);
    slot_owner_[slot] = kf;
    slot_use_[slot] = ++tick_;
    slots_[kf] = slot;
    depth_on_device_[kf] = 0;
    return slot;
}

void ProbabilityMapping::Forget(sdm::KeyFrame* kf)
{
    auto it = slots_.find(kf);
    if (it == slots_.end()) return;
    slot_owner_[it->second] = nullptr;
    slot_use_[it->second] = 0;
    depth_on_device_.erase(kf);
    map_lambdaG_.erase(kf);
    slots_.erase(it);
}

void ProbabilityMapping::InvalidateDepth(sdm::KeyFrame* kf)
{
    auto it = depth_on_device_.find(kf);
    if (it != depth_on_device_.end()) it->second = 0;
}

float ProbabilityMapping::CurrentLambdaG() const
{
    sdm_params p;
    if (sdm_get_params(ctx_, &p) != SDM_OK) return std::nanf("");
    return p.lambdaG;
}

// make the slot's depth map equal to the keyframe's host maps
void ProbabilityMapping::PushDepth(sdm::KeyFrame* kf, int slot)
{
    if (depth_on_device_[kf]) return;
    if (sdm_upload_depth(ctx_, slot, kf->depth_map_.ptr(), kf->depth_sigma_.ptr()) != SDM_OK)
        report("PushDepth");
    else
        depth_on_device_[kf] = 1;
}

// PM.cc:151-160 / 273-282: the first covisN covisible keyframes that are good and mapped
std::vector<sdm::KeyFrame*> ProbabilityMapping::PickNeighbours(sdm::KeyFrame* kf) { return PickNeighboursN(kf, opt_.covisN); }

std::vector<sdm::KeyFrame*> ProbabilityMapping::PickNeighboursN(sdm::KeyFrame* kf, int covisN)
{
    std::vector<sdm::KeyFrame*> out;
    const std::vector<sdm::KeyFrame*>& all = kf->GetVectorCovisibleKeyFrames();
    for (size_t i = 0; i < all.size(); i++) {
        if ((int)out.size() >= covisN) break;
        if (all[i]->isBad()) continue;
        if (!all[i]->Mapped()) continue;
        out.push_back(all[i]);
    }
    return out;
}

// PM.cc:370-383
void ProbabilityMapping::StereoSearchConstraints(sdm::KeyFrame* kf, float* min_depth, float* max_depth)
{
    const std::vector<float>& d = kf->GetAllPointDepths();
    if (sdm_stereo_search_constraints(d.data(), (int)d.size(), min_depth, max_depth) != SDM_OK)
        report("StereoSearchConstraints");
}

// PM.cc:137-256 (per-keyframe reconstruction) followed by PM.cc:262-315 (inter-keyframe checking of
// every keyframe whose neighbours are all reconstructed).
void ProbabilityMapping::SemiDenseRecon(sdm::KeyFrame* kf)
{
    if (!kf || kf->isBad() || kf->semidense_flag_) return;  // PM.cc:141
    std::vector<sdm::KeyFrame*> nbrs = PickNeighbours(kf);
    if ((int)nbrs.size() < opt_.covisN) return;  // PM.cc:160
    const int ref = SlotOf(kf);
    if (ref < 0) return;
    std::vector<int> nslots;
    std::vector<float> rot;
    for (size_t j = 0; j < nbrs.size(); j++) {
        int s = SlotOf(nbrs[j]);
        if (s < 0) return;
        nslots.push_back(s);
        // PM.cc:170-179: median in-plane rotation over shared map points, 0 without covisibility
        rot.push_back(sdm_median_rot_in_plane(kf->map_point_ids.data(), kf->keypoint_angles.data(),
                                              (int)std::min(kf->map_point_ids.size(), kf->keypoint_angles.size()),
                                              nbrs[j]->map_point_ids.data(), nbrs[j]->keypoint_angles.data(),
                                              (int)std::min(nbrs[j]->map_point_ids.size(),
                                                            nbrs[j]->keypoint_angles.size())));
    }
    slot_use_[ref] = ++tick_;  // keep the reference resident while its neighbours are uploaded
    float min_depth = 0.f, max_depth = 0.f;
    StereoSearchConstraints(kf, &min_depth, &max_depth);  // PM.cc:184
    if (sdm_recon(ctx_, 1, &ref, (int)nslots.size(), nslots.data(), rot.data(), &min_depth, &max_depth) != SDM_OK) {
        report("SemiDenseRecon");
        return;
    }
    if (sdm_download_depth(ctx_, ref, kf->depth_map_.ptr(), kf->depth_sigma_.ptr()) != SDM_OK) {
        report("SemiDenseRecon");
        return;
    }
    depth_on_device_[kf] = 1;
    map_lambdaG_[kf] = CurrentLambdaG();
    kf->semidense_flag_ = true;  // PM.cc:244

    // PM.cc:262-315
    if (!mpMap) return;
    std::vector<sdm::KeyFrame*> vpKFs = mpMap->GetAllKeyFrames();
    for (size_t i = 0; i < vpKFs.size(); i++) {
        sdm::KeyFrame* k = vpKFs[i];
        if (k->isBad() || k->interKF_depth_flag_) continue;
        if (!k->semidense_flag_) continue;
        std::vector<sdm::KeyFrame*> neighbors = PickNeighbours(k);
        if ((int)neighbors.size() < opt_.covisN) continue;
        int num_depth_kf = 0;
        for (size_t j = 0; j < neighbors.size(); j++)
            if (neighbors[j]->semidense_flag_) num_depth_kf++;
        if (num_depth_kf < opt_.covisN) continue;  // PM.cc:298
        InterKeyFrameDepthChecking(k, neighbors);
        UpdateSemiDensePointSet(k);
        k->interKF_depth_flag_ = true;  // PM.cc:306
    }
}

// PM.cc:628-799, in place on currentKf->depth_map_
void ProbabilityMapping::InterKeyFrameDepthChecking(sdm::KeyFrame* currentKf, std::vector<sdm::KeyFrame*> neighbors)
{
    if (!currentKf || neighbors.empty()) return;
    const int ref = SlotOf(currentKf);
    if (ref < 0) return;
    std::vector<int> nslots;
    for (size_t j = 0; j < neighbors.size(); j++) {
        int s = SlotOf(neighbors[j]);
        if (s < 0) return;
        PushDepth(neighbors[j], s);
        nslots.push_back(s);
    }
    PushDepth(currentKf, ref);
    if (sdm_inter_check(ctx_, 1, &ref, (int)nslots.size(), nslots.data(), /*commit=*/1) != SDM_OK ||
        sdm_download_depth(ctx_, ref, currentKf->depth_map_.ptr(), nullptr) != SDM_OK)
        report("InterKeyFrameDepthChecking");
}

// PM.cc:337-367
void ProbabilityMapping::UpdateSemiDensePointSet(sdm::KeyFrame* kf)
{
    const int slot = SlotOf(kf);
    if (slot < 0) return;
    PushDepth(kf, slot);
    if (sdm_pointset(ctx_, 1, &slot, /*source = depth map*/ 0) != SDM_OK ||
        sdm_download_pointset(ctx_, slot, kf->SemiDensePointSets_.ptr()) != SDM_OK)
        report("UpdateSemiDensePointSet");
}

// PM.cc:321-334
void ProbabilityMapping::UpdateAllSemiDensePointSet()
{
    if (!mpMap) return;
    std::vector<sdm::KeyFrame*> vpKFs = mpMap->GetAllKeyFrames();
    if (vpKFs.size() < 10) return;
    for (size_t i = 0; i < vpKFs.size(); i++) {
        sdm::KeyFrame* kf = vpKFs[i];
        if (kf->isBad() || !kf->interKF_depth_flag_) continue;
        if (kf->poseChanged) {
            UpdateSemiDensePointSet(kf);
            kf->poseChanged = false;
        }
    }
}

// PM.cc:385-465.  `pixel`, `F12` and `th_pi` are, at the reference's only call site (PM.cc:213-214), im_(y,x),
// ComputeFundamental(kf1,kf2) and GradTheta(y,x) of the same keyframes.  The engine derives all three from the resident
// keyframes, so a caller must pass exactly those: the arguments are CHECKED against the resident values and a mismatch is
// reported on cerr (the search then still runs on the resident values -- it cannot do anything else -- so the caller knows
// the answer is not the one the reference would give for its arguments).
void ProbabilityMapping::EpipolarSearch(sdm::KeyFrame* kf1, sdm::KeyFrame* kf2, const int x, const int y, float pixel,
                                        float min_depth, float max_depth, depthHo* dh, const float F12[9],
                                        float& best_u, float& best_v, float th_pi, float rot)
{
    const int s1 = SlotOf(kf1), s2 = SlotOf(kf2);
    if (s1 < 0 || s2 < 0 || !dh) return;
    if (x >= 0 && y >= 0 && x < kf1->im_.cols && y < kf1->im_.rows) {
        const char* what = nullptr;
        if (pixel != (float)kf1->im_.at(y, x)) what = "pixel != kf1->im_(y,x)";
        if (!kf1->GradTheta.empty() && th_pi != kf1->GradTheta.at(y, x)) what = "th_pi != kf1->GradTheta(y,x)";
        if (F12) {
            float F[9];
            if (sdm_pair_geometry(ctx_, s1, s2, F, nullptr, nullptr) == SDM_OK && memcmp(F, F12, sizeof(F)) != 0)
                what = "F12 != ComputeFundamental(kf1, kf2)";
        }
        if (what)
            std::cerr << "ProbabilityMapping::EpipolarSearch: " << what << " -- the engine searches with the resident "
                      << "keyframes' values (PM.cc:213-214 passes exactly those); this call's result differs from the "
                      << "reference's for the given arguments" << std::endl;
    }
    float out[5];
    if (sdm_epipolar_search(ctx_, s1, s2, x, y, min_depth, max_depth, rot, out) != SDM_OK) {
        report("EpipolarSearch");
        return;
    }
    if (out[2] != 0.f) {  // the reference leaves *dh untouched when no match is found
        dh->depth = out[0];
        dh->sigma = out[1];
        dh->supported = true;
        best_u = out[3];
        best_v = out[4];
    }
}

// PM.cc:877-910
void ProbabilityMapping::GetSearchRange(float& umin, float& umax, int px, int py, float mind, float maxd,
                                        sdm::KeyFrame* kf, sdm::KeyFrame* kf2)
{
    const int s1 = SlotOf(kf), s2 = SlotOf(kf2);
    if (s1 < 0 || s2 < 0) return;
    if (sdm_search_range(ctx_, s1, s2, px, py, mind, maxd, &umin, &umax) != SDM_OK) report("GetSearchRange");
}

// PM.cc:598-626
void ProbabilityMapping::InverseDepthHypothesisFusion(const std::vector<depthHo>& h, depthHo& dist)
{
    dist.depth = 0;
    dist.sigma = 0;
    dist.supported = false;
    if (!ctx_) {
        std::cerr << "ProbabilityMapping::InverseDepthHypothesisFusion: no device context yet" << std::endl;
        return;
    }
    std::vector<float> rho(h.size()), sig(h.size());
    for (size_t i = 0; i < h.size(); i++) {
        rho[i] = h[i].depth;
        sig[i] = h[i].sigma;
    }
    float out[3];
    if (sdm_fuse(ctx_, rho.data(), sig.data(), (int)h.size(), out) != SDM_OK) {
        report("InverseDepthHypothesisFusion");
        return;
    }
    if (out[2] != 0.f) {
        dist.depth = out[0];
        dist.sigma = out[1];
        dist.supported = true;
    }
}

// PM.cc:486-547
void ProbabilityMapping::IntraKeyFrameDepthChecking(sdm::Mat<float>& depth_map, sdm::Mat<float>& depth_sigma,
                                                    const sdm::Mat<float> gradimg)
{
    if (!Ensure(depth_map.cols, depth_map.rows)) return;
    if (sdm_intra_check_maps(ctx_, depth_map.ptr(), depth_sigma.ptr(), gradimg.empty() ? nullptr : gradimg.ptr()) !=
        SDM_OK)
        report("IntraKeyFrameDepthChecking");
}

// PM.cc:549-596
void ProbabilityMapping::IntraKeyFrameDepthGrowing(sdm::Mat<float>& depth_map, sdm::Mat<float>& depth_sigma,
                                                   const sdm::Mat<float> gradimg)
{
    if (!Ensure(depth_map.cols, depth_map.rows)) return;
    if (sdm_intra_grow_maps(ctx_, depth_map.ptr(), depth_sigma.ptr(), gradimg.ptr()) != SDM_OK)
        report("IntraKeyFrameDepthGrowing");
}

// PM.cc:972-986
void ProbabilityMapping::ComputeFundamental(sdm::KeyFrame* pKF1, sdm::KeyFrame* pKF2, float F12[9])
{
    const int s1 = SlotOf(pKF1), s2 = SlotOf(pKF2);
    if (s1 < 0 || s2 < 0) return;
    if (sdm_pair_geometry(ctx_, s1, s2, F12, nullptr, nullptr) != SDM_OK) report("ComputeFundamental");
}

// PM.cc:100-132
long ProbabilityMapping::SavePointCloudObj(const char* path)
{
    std::ofstream out(path, std::ios::out);
    if (!out) {
        std::cerr << "Failed to save points on line" << std::endl;
        return -1;
    }
    long n = 0;
    if (!mpMap) return 0;
    std::vector<sdm::KeyFrame*> vpKFs = mpMap->GetAllKeyFrames();
    for (size_t i = 0; i < vpKFs.size(); i++) {
        sdm::KeyFrame* kf = vpKFs[i];
        if (kf->isBad() || !kf->semidense_flag_ || !kf->interKF_depth_flag_) continue;
        for (int y = 0; y < kf->im_.rows; y++)
            for (int x = 0; x < kf->im_.cols; x++) {
                if (kf->depth_sigma_.at(y, x) > 0.01) continue;
                if (kf->depth_map_.at(y, x) > 0.000001) {
                    out << "v " + std::to_string(kf->SemiDensePointSets_.at(y, 3 * x)) + " " +
                               std::to_string(kf->SemiDensePointSets_.at(y, 3 * x + 1)) + " " +
                               std::to_string(kf->SemiDensePointSets_.at(y, 3 * x + 2))
                        << std::endl;
                    n++;
                }
            }
    }
    out.flush();
    return n;
}

// SFMTranscriptInterface_ORBSLAM.cpp:319-374 (addKeyFrameInsertionWithLinesEntry): text form only.
long ProbabilityMapping::AppendTranscriptEntry(sdm::KeyFrame* kf, int camIndex, int camIndexOriginal, std::ostream& out,
                                               double max_sigma)
{
    if (!kf || kf->SemiDensePointSets_.empty()) return -1;
    // camera centre = translation of Twc = -Rcw^T * tcw (src/KeyFrame.cc:70-84), float, left-to-right sums
    float Ow[3];
    for (int i = 0; i < 3; i++) {
        float p0 = kf->Tcw[0 * 4 + i] * kf->Tcw[3];
        float p1 = kf->Tcw[1 * 4 + i] * kf->Tcw[7];
        float p2 = kf->Tcw[2 * 4 + i] * kf->Tcw[11];
        Ow[i] = -((p0 + p1) + p2);
    }
    out << "new cam: [" << (double)Ow[0] << "; " << (double)Ow[1] << "; " << (double)Ow[2] << "] {" << std::endl;
    long n = 0;
    for (int y = 0; y < kf->im_.rows; y++)
        for (int x = 0; x < kf->im_.cols; x++) {
            if (kf->depth_sigma_.at(y, x) > max_sigma) continue;  // PM.cc:120 (0.01)
            if (!(kf->depth_map_.at(y, x) > 0.000001)) continue;  // PM.cc:121
            out << "new point: [" << (double)kf->SemiDensePointSets_.at(y, 3 * x) << "; "
                << (double)kf->SemiDensePointSets_.at(y, 3 * x + 1) << "; "
                << (double)kf->SemiDensePointSets_.at(y, 3 * x + 2) << "]"
                << ", " << camIndex << ", " << camIndexOriginal << std::endl;
            n++;
        }
    out << "}" << std::endl;
    return n;
}

// ---- multi-GPU block driver -----------------------------------------------------------------
bool ProbabilityMapping::InitSharding(const unsigned char* comm_id, int world, int rank)
{
    if (!ctx_) {
        std::cerr << "ProbabilityMapping::InitSharding: no device context yet (upload a keyframe first or size the "
                     "context with SemiDenseReconBlock)" << std::endl;
        return false;
    }
    if (sdm_comm_init(ctx_, comm_id, world, rank) != SDM_OK) {
        report("InitSharding");
        return false;
    }
    return true;
}

// What one rank does in a sharded pass, derived identically on every rank from REPLICATED state only: the keyframe
// list, the covisibility lists and the flags bad / mapped / semidense_flag_ / interKF_depth_flag_ (SemiDenseReconBlock
// keeps the two stage flags replicated: after a pass it sets them on every keyframe that ANY rank reconstructed or
// checked, see the end of that function).  The C++ counterpart of shard.py's plan().  Blocks are equal and contiguous:
// owner(i) = i / count.
//   refs   keyframes reconstructed in this pass (PM.cc:141,157,160: good, mapped, not yet semidense, covisN good+mapped
//          neighbours)
//   check  keyframes checked in this pass (PM.cc:265-298: good, not yet checked, semidense by the end of this pass's
//          reconstruction, covisN neighbours that are ALL semidense by then) -- the reference's gate: a keyframe whose
//          neighbour has no depth map yet is not checked, and nobody sends or receives a map that does not exist
bool ProbabilityMapping::PlanBlock(const std::vector<sdm::KeyFrame*>& all, int first, int count, int world, int rank,
                                   int covisN, sdm::BlockPlan* out)
{
    const int n_all = (int)all.size();
    if (!out || count < 1 || first < 0 || first + count > n_all || world < 1 || rank < 0 || rank >= world) return false;
    if (world > 1 && (n_all != world * count || first != rank * count)) return false;
    std::map<sdm::KeyFrame*, int> index;
    for (int i = 0; i < n_all; i++) index[all[i]] = i;
    // PM.cc:151-160 as indices; empty = fewer than covisN usable neighbours (or one outside `all`)
    auto nbr_idx = [&](int i) {
        std::vector<int> r;
        std::vector<sdm::KeyFrame*> nb = PickNeighboursN(all[i], covisN);
        if ((int)nb.size() < covisN) return r;
        for (sdm::KeyFrame* p : nb) {
            auto it = index.find(p);
            if (it == index.end()) return std::vector<int>();
            r.push_back(it->second);
        }
        return r;
    };
    *out = sdm::BlockPlan();
    out->needed.assign(n_all, 0);
    out->boundary.assign(n_all, 0);
    out->recon_all.assign(n_all, 0);
    out->check_all.assign(n_all, 0);
    std::vector<std::vector<int> > nb_all(n_all);
    for (int i = 0; i < n_all; i++) {
        if (all[i]->isBad()) continue;
        nb_all[i] = nbr_idx(i);
        if (!all[i]->semidense_flag_ && all[i]->Mapped() && !nb_all[i].empty()) out->recon_all[i] = 1;  // PM.cc:141,157,160
    }
    auto will_have_map = [&](int j) { return all[j]->semidense_flag_ || out->recon_all[j]; };
    for (int i = 0; i < n_all; i++) {
        if (all[i]->isBad() || all[i]->interKF_depth_flag_ || nb_all[i].empty() || !will_have_map(i)) continue;
        bool ready = true;  // PM.cc:292-298
        for (int j : nb_all[i]) ready = ready && will_have_map(j);
        if (ready) out->check_all[i] = 1;
    }
    for (int i = first; i < first + count; i++) {
        out->needed[i] = 1;
        if (out->recon_all[i]) {
            out->refs.push_back(i);
            out->nbrs.push_back(nb_all[i]);
            for (int j : nb_all[i]) out->needed[j] = 1;
        }
        if (out->check_all[i]) {
            out->check.push_back(i);
            out->check_nbrs.push_back(nb_all[i]);
            for (int j : nb_all[i]) out->needed[j] = 1;
        }
    }
    if (world == 1) return true;
    for (int q = 0; q < world; q++) {
        if (q == rank) continue;
        std::vector<char> wanted(n_all, 0);
        for (int i = q * count; i < (q + 1) * count; i++)
            if (out->check_all[i])
                for (int j : nb_all[i])
                    if (j >= first && j < first + count) wanted[j] = 1;
        for (int j = first; j < first + count; j++)
            if (wanted[j]) {
                out->send_peer.push_back(q);
                out->send_kf.push_back(j);
                out->boundary[j] = 1;
            }
    }
    std::vector<char> fetch(n_all, 0);
    for (const std::vector<int>& row : out->check_nbrs)
        for (int j : row)
            if (!(j >= first && j < first + count)) fetch[j] = 1;
    for (int j = 0; j < n_all; j++)
        if (fetch[j]) {
            out->recv_peer.push_back(j / count);
            out->recv_kf.push_back(j);
        }
    return true;
}

bool ProbabilityMapping::CompactSourcesReady(const std::vector<sdm::KeyFrame*>& kfs)
{
    if (!ctx_) return false;
    std::vector<int> s;
    for (sdm::KeyFrame* kf : kfs) {
        auto it = slots_.find(kf);
        if (it == slots_.end()) return false;
        s.push_back(it->second);
    }
    int ready = 0;
    if (sdm_compact_sources_ready(ctx_, (int)s.size(), s.data(), &ready) != SDM_OK) {
        report("CompactSourcesReady");
        return false;
    }
    return ready != 0;
}

void ProbabilityMapping::SemiDenseReconBlock(const std::vector<sdm::KeyFrame*>& all, int first, int count)
{
    const int n_all = (int)all.size();
    if (count < 1 || first < 0 || first + count > n_all) {
        std::cerr << "ProbabilityMapping::SemiDenseReconBlock: block out of range" << std::endl;
        return;
    }
    if (!Ensure(all[first]->im_.cols, all[first]->im_.rows)) return;
    int world = 1, rank = 0;
    sdm_comm_info(ctx_, &world, &rank);
    sdm::BlockPlan plan;
    if (!PlanBlock(all, first, count, world, rank, opt_.covisN, &plan)) {
        std::cerr << "ProbabilityMapping::SemiDenseReconBlock: blocks must be equal and contiguous (rank*count)" << std::endl;
        return;
    }
    const std::vector<int>& refs = plan.refs;
    const std::vector<std::vector<int> >& nbrs = plan.nbrs;
    const std::vector<char>& needed = plan.needed;
    const std::vector<char>& is_boundary = plan.boundary;

    // ---- everything that can fail locally happens BEFORE the first transfer is posted, and the ranks agree on a
    // go / no-go: a rank that returned early here used to leave its peers waiting for maps that never came
    bool local_ok = true;
    int n_needed = 0;
    for (int i = 0; i < n_all; i++) n_needed += needed[i];
    if (n_needed > (int)slot_owner_.size()) {
        std::cerr << "ProbabilityMapping::SemiDenseReconBlock: " << n_needed << " keyframes (block + covisible halo) "
                  << "exceed Options::max_keyframes = " << slot_owner_.size() << std::endl;
        local_ok = false;
    }
    std::vector<int> slot(n_all, -1);
    for (int i = 0; i < n_all && local_ok; i++)
        if (needed[i] && (slot[i] = SlotOf(all[i])) < 0) local_ok = false;
    if (local_ok) {
        // own keyframes reconstructed in an earlier pass answer with the map they have.  While the device slot still holds
        // that map nothing is touched (it stays a pipeline map the compact exchange accepts); a slot that was recycled gets
        // the host copy back -- the reconstructed or the checked map, both zero outside the keyframe's pixel list
        // (PM.cc:201, 662) -- and is declared a pipeline map again, PROVIDED it was reconstructed under the lambdaG in force
        // now: under another threshold its support can lie outside the list the declaration rebuilds, and the list kernels and
        // the compact send would drop those pixels.  Such a map stays an ordinary map; the pass then moves whole maps (the
        // compact-source question below answers no).  A keyframe without a map is never marked: nobody's plan reads it
        // (PlanBlock)
        const float lam_now = CurrentLambdaG();
        for (int i = first; i < first + count && local_ok; i++) {
            if (!all[i]->semidense_flag_ || depth_on_device_[all[i]]) continue;
            PushDepth(all[i], slot[i]);
            if (!depth_on_device_[all[i]]) {
                report("SemiDenseReconBlock");
                local_ok = false;
                continue;
            }
            auto it = map_lambdaG_.find(all[i]);
            if (it != map_lambdaG_.end() && it->second == lam_now && sdm_assume_pipeline_maps(ctx_, 1, &slot[i]) != SDM_OK) {
                report("SemiDenseReconBlock");
                local_ok = false;
            }
        }
    }
    int all_ok = 0;
    if (sdm_comm_all_ok(ctx_, local_ok ? 1 : 0, &all_ok) != SDM_OK) {
        report("SemiDenseReconBlock");
        return;
    }
    if (!all_ok) {
        if (local_ok) std::cerr << "ProbabilityMapping::SemiDenseReconBlock: another rank cannot run this pass; skipped" << std::endl;
        return;
    }
    if (world > 1) {
        // wire format of this pass: the longest active list among the keyframes ANY rank touches, rounded up to 64 entries
        // (the same on every rank by construction: one all-reduce); whole maps if that is more than half a plane
        int longest = 0, all_longest = 0, entries = 0;
        for (int i = 0; i < n_all && opt_.exchange_compact; i++) {
            int cnt = 0;
            if (needed[i] && sdm_active_count(ctx_, slot[i], &cnt) == SDM_OK) longest = std::max(longest, cnt);
        }
        if (opt_.exchange_compact) {
            // ... and every map this rank sends must qualify as a compact source: the ones it reconstructs in this pass do
            // by construction, the ones from earlier passes are asked about.  A rank that cannot says so through the same
            // all-reduce (a "longest list" no wire format can hold): ALL ranks then move whole maps in this pass.
            std::vector<int> old_src;
            for (int j : plan.send_kf)
                if (!plan.recon_all[j]) old_src.push_back(slot[j]);
            int ready = 1;
            if (!old_src.empty() && sdm_compact_sources_ready(ctx_, (int)old_src.size(), old_src.data(), &ready) != SDM_OK) ready = 0;
            if (!ready) longest = std::numeric_limits<int>::max() / 2;
        }
        if (sdm_comm_all_max(ctx_, longest, &all_longest) != SDM_OK) {
            report("SemiDenseReconBlock");
            return;
        }
        if (opt_.exchange_compact) {
            const long long P = (long long)all[first]->im_.cols * all[first]->im_.rows;
            const long long e = ((long long)all_longest + 63) / 64 * 64;
            entries = (e * 2 > P) ? 0 : (int)e;
        }
        if (sdm_exchange_compact(ctx_, entries) != SDM_OK) {
            report("SemiDenseReconBlock");
            return;
        }
    }

    std::vector<int> send_peer = plan.send_peer, recv_peer = plan.recv_peer, send_slot, recv_slot;
    for (int j : plan.send_kf) send_slot.push_back(slot[j]);
    for (int j : plan.recv_kf) recv_slot.push_back(slot[j]);
    // per-reference constants: depth prior (PM.cc:184) and median in-plane rotations (PM.cc:170-179)
    const int n = opt_.covisN;
    auto run_recon = [&](bool boundary) {
        std::vector<int> r, ns;
        std::vector<float> rot, mind, maxd;
        for (size_t a = 0; a < refs.size(); a++) {
            if ((is_boundary[refs[a]] != 0) != boundary) continue;
            sdm::KeyFrame* kf = all[refs[a]];
            r.push_back(slot[refs[a]]);
            float mn = 0.f, mx = 0.f;
            StereoSearchConstraints(kf, &mn, &mx);
            mind.push_back(mn);
            maxd.push_back(mx);
            for (int j : nbrs[a]) {
                sdm::KeyFrame* k2 = all[j];
                ns.push_back(slot[j]);
                rot.push_back(sdm_median_rot_in_plane(
                    kf->map_point_ids.data(), kf->keypoint_angles.data(),
                    (int)std::min(kf->map_point_ids.size(), kf->keypoint_angles.size()), k2->map_point_ids.data(),
                    k2->keypoint_angles.data(), (int)std::min(k2->map_point_ids.size(), k2->keypoint_angles.size())));
            }
        }
        if (r.empty()) return true;
        if (sdm_recon(ctx_, (int)r.size(), r.data(), n, ns.data(), rot.data(), mind.data(), maxd.data()) != SDM_OK) {
            report("SemiDenseReconBlock");
            return false;
        }
        return true;
    };
    // From here to sdm_exchange_wait this rank takes part in the exchange WHATEVER happens locally: a failed
    // reconstruction still posts the planned sends (of whatever the slots hold) and receives, and is reported after
    bool ok = run_recon(true);  // the keyframes other ranks read first ...
    // (has_depth + "pipeline map" keep the compact send's own precondition check from failing on this rank alone; what
    // the slots hold is sent, and the pass is voided for every rank by the agreement at its end)
    if (!ok && !send_slot.empty()) (void)sdm_assume_pipeline_maps(ctx_, (int)send_slot.size(), send_slot.data());
    if (sdm_exchange_halo_begin(ctx_, (int)send_peer.size(), send_peer.data(), send_slot.data(), (int)recv_peer.size(),
                                recv_peer.data(), recv_slot.data()) != SDM_OK) {
        report("SemiDenseReconBlock");  // argument / state errors are the same on every rank (replicated plan); an RCCL
        ok = false;                     // failure is fatal for the communicator anyway
    }
    ok = run_recon(false) && ok;  // ... the rest while those maps travel
    // PM.cc:300-306 for every keyframe of the block that is ready, snapshot order (the pool keeps the maps as
    // reconstructed: commit = 0).  Keyframes whose neighbours are all this rank's own are checked BEFORE the wait -- more
    // work for the transfer to hide behind -- the others after it.
    std::vector<int> check_slot(plan.check.size(), -1);
    auto run_check = [&](bool remote) {
        std::vector<int> r, ns;
        for (size_t a = 0; a < plan.check.size(); a++) {
            bool reads_remote = false;
            for (int j : plan.check_nbrs[a]) reads_remote = reads_remote || j < first || j >= first + count;
            if (reads_remote != remote) continue;
            check_slot[a] = slot[plan.check[a]];
            r.push_back(check_slot[a]);
            for (int j : plan.check_nbrs[a]) ns.push_back(slot[j]);
        }
        if (r.empty()) return true;
        if (sdm_inter_check_pointset(ctx_, (int)r.size(), r.data(), n, ns.data(), /*commit=*/0) != SDM_OK) {
            report("SemiDenseReconBlock");
            return false;
        }
        return true;
    };
    bool checked = ok && run_check(false);
    if (sdm_exchange_wait(ctx_) != SDM_OK) {
        report("SemiDenseReconBlock");
        ok = false;
    }
    checked = ok && checked && run_check(true);
    // a compact map whose list differs from this rank's list of that keyframe was refused by the receiver (its plane is
    // not the peer's map): the checks that read it are void
    int refused = 0;
    if (world > 1 && sdm_exchange_mismatches(ctx_, &refused) != SDM_OK) {
        report("SemiDenseReconBlock");
        ok = false;
    }
    if (refused > 0) {
        std::cerr << "ProbabilityMapping::SemiDenseReconBlock: " << refused << " received map(s) refused (the sender's pixel "
                  << "list differs from this rank's): pass voided" << std::endl;
        ok = false;
    }
    // ---- the pass counts for everybody or for nobody: the stage flags are REPLICATED state (the next pass's plans are
    // derived from them on every rank and must pair up), so they are set only when every rank finished its share; after a
    // local failure anywhere they stay as they were on ALL ranks and the pass can be repeated
    int pass_ok = 0;
    if (sdm_comm_all_ok(ctx_, (ok && checked) ? 1 : 0, &pass_ok) != SDM_OK) {
        report("SemiDenseReconBlock");
        return;
    }
    if (!pass_ok) {
        if (ok && checked)
            std::cerr << "ProbabilityMapping::SemiDenseReconBlock: another rank failed in this pass; no flags set" << std::endl;
        // the device maps this pass wrote are not the keyframes' maps: the next pass starts from the host copies
        for (size_t a = 0; a < refs.size(); a++) depth_on_device_[all[refs[a]]] = 0;
        return;
    }
    // the maps as SemiDenseRecon left them (PM.cc:244)
    for (size_t a = 0; a < refs.size(); a++) {
        sdm::KeyFrame* kf = all[refs[a]];
        if (sdm_download_depth(ctx_, slot[refs[a]], kf->depth_map_.ptr(), kf->depth_sigma_.ptr()) != SDM_OK) report("SemiDenseReconBlock");
        depth_on_device_[kf] = 1;
        map_lambdaG_[kf] = CurrentLambdaG();
        kf->semidense_flag_ = true;
    }
    for (size_t a = 0; a < plan.check.size(); a++) {
        sdm::KeyFrame* kf = all[plan.check[a]];
        if (sdm_download_checked(ctx_, check_slot[a], kf->depth_map_.ptr()) != SDM_OK ||
            sdm_download_pointset(ctx_, check_slot[a], kf->SemiDensePointSets_.ptr()) != SDM_OK)
            report("SemiDenseReconBlock");
        depth_on_device_[kf] = 0;  // the host map is now the CHECKED one; the device pool still holds the snapshot
        kf->interKF_depth_flag_ = true;  // PM.cc:306
    }
    // what the other ranks reconstructed / checked in this pass (their maps live on their GPUs; this rank's copies of
    // those keyframes carry only the flags)
    for (int i = 0; i < n_all; i++) {
        if (plan.recon_all[i]) all[i]->semidense_flag_ = true;
        if (plan.check_all[i]) all[i]->interKF_depth_flag_ = true;
    }
}
