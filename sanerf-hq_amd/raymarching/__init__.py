from .raymarching import (  # noqa: F401
    composite, contract, distort_loss, generate_rays, rays_from_pixels, grid_composite, mask_head, mask_head_fusable, mask_nll, mlp_forward, near_far_from_aabb, proposal_loss_stage, render_rays, sample_pdf, sample_positions, weights_from_sigma, jitter, ray_composite, proposal_loss_all, zeros_f32,
    RenderPlan, Tuning, tuning, last_launch_info, PROPOSAL_LOSS_MAX_T, WEIGHTS_BACKWARD_MAX_T, DISTORT_LOSS_MAX_T, FP16_SPLIT_LIMIT, mlp_wide_overflow, _host_values,
)
