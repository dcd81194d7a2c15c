"""Host-side mirror of PromptEnhancerService (server-node/src/services/promptEnhancer.js).

Pure string work that stays on the host (step 2).  Same rules as the reference:
threshold 0.3 (strict), stable sort by confidence descending, top 3, severity bands 0.7 / 0.5,
template lookup, parts joined by one space, 1000 -> 950 + '...' truncation.  The template texts
are the reference's prompt vocabulary (promptEnhancer.js:9-51) -- data the provider seam expects,
kept verbatim so the enhanced prompt is byte-identical for identical scores.
"""

KEYS = ["blur", "noise", "lowLight", "compression", "scratch", "fade", "colorShift"]

# promptEnhancer.js:9-45
DEGRADATION_TEMPLATES = {
    "blur": {"high": "reduce severe motion blur and sharpen edges while preserving natural detail",
             "medium": "reduce motion blur and improve focus clarity",
             "low": "slightly enhance sharpness and edge definition"},
    "noise": {"high": "aggressively suppress grain and noise while preserving fine detail and texture",
              "medium": "reduce noise and grain while maintaining image detail",
              "low": "lightly reduce noise without affecting texture"},
    "lowLight": {"high": "significantly enhance brightness and recover shadow detail without overexposure",
                 "medium": "improve brightness and enhance shadow areas",
                 "low": "slightly brighten dark areas and improve visibility"},
    "compression": {"high": "remove severe JPEG artifacts and restore texture quality",
                    "medium": "reduce compression artifacts and improve image quality",
                    "low": "minimize minor compression artifacts"},
    "scratch": {"high": "remove scratches, blemishes, and physical damage using advanced inpainting",
                "medium": "repair visible scratches and minor damage",
                "low": "touch up small blemishes and imperfections"},
    "fade": {"high": "restore vibrant colors and dramatically improve contrast",
             "medium": "enhance color vibrancy and increase contrast",
             "low": "slightly boost colors and improve contrast"},
    "colorShift": {"high": "correct severe color cast and restore natural white balance",
                   "medium": "adjust color balance and improve white balance",
                   "low": "fine-tune color balance for natural appearance"},
}

# promptEnhancer.js:47-51
BASE_INSTRUCTIONS = {
    "quality": "Maintain the highest possible image quality and preserve important details",
    "naturalness": "Ensure the result looks natural and realistic, avoiding over-processing",
    "preservation": "Preserve the original composition, subject matter, and artistic intent",
}


def determine_severity(confidence):
    """promptEnhancer.js:141-145"""
    if confidence >= 0.7:
        return "high"
    if confidence >= 0.5:
        return "medium"
    return "low"


def identify_top_issues(degradation):
    """promptEnhancer.js:121-136 -- insertion order of the dict is the tie-break (stable sort)."""
    issues = []
    for type_, confidence in degradation.items():
        if confidence > 0.3:
            issues.append({"type": type_, "confidence": confidence, "severity": determine_severity(confidence)})
    issues.sort(key=lambda i: -i["confidence"])  # Python's sort is stable like V8's
    return issues[:3]


def argmax_label(degradation):
    """label = keys[argmax(scores)], first max wins in key order."""
    best = None
    for k in KEYS:
        if best is None or degradation[k] > degradation[best]:
            best = k
    return best


class PromptEnhancerService:
    def __init__(self, logger=None):
        self.logger = logger

    def _warn(self, msg, **kw):
        if self.logger is not None:
            self.logger.warning("%s %s", msg, kw)

    def _generate_degradation_instructions(self, issues):
        """promptEnhancer.js:150-160"""
        out = []
        for issue in issues:
            template = DEGRADATION_TEMPLATES.get(issue["type"])
            if not template:
                self._warn(f"[prompt-enhancer] No template for degradation type: {issue['type']}")
                out.append(f"address {issue['type']} issues")
            else:
                out.append(template.get(issue["severity"]) or template["medium"])
        return out

    def _build_prompt(self, user_prompt, instructions, issues):
        """promptEnhancer.js:165-205"""
        parts = []
        if user_prompt and user_prompt.strip():
            parts.append(f"User request: {user_prompt.strip()}.")
        if instructions:
            parts.append(f"Technical restoration: {', '.join(instructions)}.")
        quality = ", ".join([BASE_INSTRUCTIONS["quality"], BASE_INSTRUCTIONS["naturalness"],
                             BASE_INSTRUCTIONS["preservation"]])
        parts.append(f"Quality guidelines: {quality}.")
        if any(i["severity"] == "high" for i in issues):
            parts.append("This image requires significant restoration work - apply corrections carefully to avoid artifacts.")
        elif len(issues) == 0:
            parts.append("This image appears to be in good condition - apply subtle enhancements only.")
        prompt = " ".join(parts)
        if len(prompt) > 1000:
            prompt = prompt[:950] + "..."
            self._warn("[prompt-enhancer] Prompt truncated due to length")
        return prompt

    def enhance(self, degradation, user_prompt=None, options=None):
        """promptEnhancer.js:65-116 (options is accepted and unused, as in the reference)."""
        issues = identify_top_issues(degradation)
        return self._build_prompt(user_prompt, self._generate_degradation_instructions(issues), issues)

    @staticmethod
    def validate_degradation(degradation):
        """promptEnhancer.js:217-232"""
        for t in KEYS:
            if t not in degradation:
                raise ValueError(f"Missing degradation type: {t}")
            v = degradation[t]
            if not isinstance(v, (int, float)) or isinstance(v, bool) or v < 0 or v > 1:
                raise ValueError(f"Invalid degradation value for {t}: must be number between 0 and 1")
        return True
